#!/usr/bin/env python
"""Round 3: tools/collect_r02.py (kernel stats, HBM bytes, SQ / TCC counters of the batched lk_residual_kernel launch) plus
  * the fp64 instruction counters of pass sq3 (SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64, per-SIMD wave instructions) -> fp64 FLOP per
    launch = (2 FMA + ADD + MUL + TRANS) x 64 lanes, per point and per wave - the roofline the kernel actually lives on (fp64 vector
    peak 78.6 TFLOP/s);
  * where the numbers come from: the commit (LK_PROF_COMMIT) and a fingerprint of the sources of the batch residual kernel
    (bench.py recomputes it and warns when the kernel has changed since the counters were taken)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r03a"
P = os.environ.get("LK_PROFILES_DIR", os.path.join(ROOT, "profiles"))
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "collect_r02.py"), tag], check=False, stdout=subprocess.DEVNULL)
out = json.load(open(os.path.join(P, "latest_pmc.json")))
att = json.load(open(os.path.join(P, f"{tag}_pmc_attrib.json")))
g = lambda k: att.get(k, {}).get("avg")   # noqa: E731
if g("SQ_INSTS_VALU_FMA_F64") is not None:
    fma, add, mul, trans = g("SQ_INSTS_VALU_FMA_F64"), g("SQ_INSTS_VALU_ADD_F64") or 0.0, g("SQ_INSTS_VALU_MUL_F64") or 0.0, g("SQ_INSTS_VALU_TRANS_F64") or 0.0
    flops = (2.0 * fma + add + mul + trans) * 64.0
    out.update({
        "fp64_insts_per_launch": {"fma": fma, "add": add, "mul": mul, "trans": trans, "int64": g("SQ_INSTS_VALU_INT64")},
        "fp64_flops_per_launch": flops, "fp64_flops_per_point": flops / out["points_per_launch"],
        "fp64_insts_per_wave": (fma + add + mul + trans) / out["waves"],
        "fp64_formula": "(2 x FMA_F64 + ADD_F64 + MUL_F64 + TRANS_F64) x 64 lanes (wave-level instruction counts; lanes masked off by EXEC are counted as executed)",
    })
import bench  # noqa: E402

out["commit"] = os.environ.get("LK_PROF_COMMIT", "unknown")
out["kernel_sources_sha16"] = bench.kernel_sources_sha16()
out["compulsory_bytes_per_point"] = 20
json.dump(out, open(os.path.join(P, "latest_pmc.json"), "w"), indent=1)
json.dump(out, open(os.path.join(P, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
