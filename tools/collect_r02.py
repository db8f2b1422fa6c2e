#!/usr/bin/env python
"""Condense the passes of tools/gpu_prof_r02.sh (merged back into gpurun_out/) into committed summaries:
  profiles/<tag>_kernel_stats.csv, <tag>_pmc_residual.json   (tools/collect_profiles.py)
  profiles/<tag>_pmc_attrib.json                              (tools/collect_pmc.py)
  profiles/latest_pmc.json   per-point / per-wave figures of the batched lk_residual_kernel launch that bench.py turns into
                             roofline.{frac, traffic, l2_frac, valu_issue_frac} with the launch time of ITS timed region.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02a"
P = os.environ.get("LK_PROFILES_DIR", os.path.join(ROOT, "profiles"))
os.makedirs(P, exist_ok=True)
for tool in ("collect_profiles.py", "collect_pmc.py"):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), tag], check=False, stdout=subprocess.DEVNULL)
res = json.load(open(os.path.join(P, f"{tag}_pmc_residual.json")))
att = json.load(open(os.path.join(P, f"{tag}_pmc_attrib.json")))
big = max(res["by_geometry"].values(), key=lambda e: e["points_upper"])
pts = big["points_upper"]
waves = att["SQ_WAVES"]["avg"]
out = {
    "tag": tag, "kernel": "lk_residual_kernel<false>", "slots": big["slots"], "points_per_launch": pts,
    "unique_scans": big["slots"],
    "hbm_bytes_per_launch": big.get("hbm_bytes_per_launch"), "hbm_bytes_per_point": big.get("hbm_bytes_per_point"),
    "FETCH_SIZE_KiB": big.get("FETCH_SIZE_KiB"), "WRITE_SIZE_KiB": big.get("WRITE_SIZE_KiB"),
    "hbm_formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 64 B per 128-B request)",
    "waves": waves, "valu_insts_per_wave": att["SQ_INSTS_VALU"]["avg"] / waves,
    "salu_insts_per_wave": att.get("SQ_INSTS_SALU", {}).get("avg", 0) / waves,
    "lds_insts_per_wave": att.get("SQ_INSTS_LDS", {}).get("avg", 0) / waves,
    "vmem_rd_insts_per_wave": att.get("SQ_INSTS_VMEM_RD", {}).get("avg", 0) / waves,
    "mfma_f64_ops": att.get("SQ_INSTS_VALU_MFMA_MOPS_F64", {}).get("avg"),
    "tcc_req_per_point": att["TCC_REQ_sum"]["avg"] / pts if "TCC_REQ_sum" in att else None,
    "tcc_hit_rate": att["TCC_HIT_sum"]["avg"] / (att["TCC_HIT_sum"]["avg"] + att["TCC_MISS_sum"]["avg"]) if "TCC_HIT_sum" in att else None,
    "wait_any_frac_of_wave_cycles": att["SQ_WAIT_ANY"]["avg"] / att["SQ_WAVE_CYCLES"]["avg"] if "SQ_WAIT_ANY" in att else None,
    "gui_active_cycles": att.get("GRBM_GUI_ACTIVE", {}).get("avg"),
    "profiled_dur_us": att["SQ_INSTS_VALU"]["avg_dur_us_profiled"],
}
if out["gui_active_cycles"]:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    out["clock_GHz_under_profiler"] = out["gui_active_cycles"] / 8.0 / (out["profiled_dur_us"] * 1e3)
json.dump(out, open(os.path.join(P, "latest_pmc.json"), "w"), indent=1)
json.dump(out, open(os.path.join(P, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
