#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/gpu_prof_config2.sh (/tmp/c2p_*) into
  <tag>_config2_kernel_stats.csv    the --stats table of the kernel-trace pass
  <tag>_config2_pmc_summary.json    raw counter sums per kernel of interest (the rows kernel + the two calibration kernels)
  latest_config2_pmc.json           what bench.py's extra.config2_roofline reads
in $LK_PROFILES_DIR (default profiles/).   usage: collect_config2_pmc.py <tag> <rows launches in the profiled run> <slots>

HBM bytes: FETCH_SIZE / WRITE_SIZE are KiB counters.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced reads (doubled here), "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count".  The workload therefore
runs a fill (N x 48 B written) and an elementwise multiply (N x 48 B read and written) in the same profiled process; their counter readings against
the known byte counts are stored as `calibration`, and the rows kernel's own write count has a known value too (it writes a 64-B row record and the valid
byte per point unconditionally)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def short(name):
    return name.replace("void ", "").split("(")[0]


def main():
    tag, launches, slots = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out = os.environ.get("LK_PROFILES_DIR", os.path.join(ROOT, "profiles"))
    os.makedirs(out, exist_ok=True)
    N = slots * bench.N_PTS
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(lambda: collections.defaultdict(set))

    def cls(name):
        if "lk_residual_kernel<true" in name or "lk_residual_kernel<(bool)1" in name:
            return "rows"
        if "FillFunctor" in name:
            return "fill"
        if "MulFunctor" in name or "mul" in name.lower() and "at::native" in name:
            return "mul"
        return None

    for f in glob.glob("/tmp/c2p_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c = cls(short(r["Kernel_Name"]))
            if c is None:
                continue
            if c in ("fill", "mul") and int(float(r.get("Grid_Size") or 0)) < N // 8:   # only the big calibration launches (the map build etc. also fill small buffers)
                continue
            acc[c][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[c][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
    dur = collections.defaultdict(list)
    for f in glob.glob("/tmp/c2p_stats/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c = cls(short(r["Kernel_Name"]))
            if c is None:
                continue
            if c in ("fill", "mul") and int(float(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)) < N // 8:
                continue
            dur[c].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for f in glob.glob("/tmp/c2p_stats/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, os.path.join(out, f"{tag}_config2_kernel_stats.csv"))
    per = {}
    for c, d in acc.items():
        per[c] = {k: v / max(1, len(ndisp[c][k])) for k, v in d.items()}   # average per dispatch
        per[c]["dispatches_counted"] = {k: len(v) for k, v in ndisp[c].items()}
        if dur.get(c):
            per[c]["avg_dur_us_trace_pass"] = sum(dur[c]) / len(dur[c])
            per[c]["dispatches_trace_pass"] = len(dur[c])
    json.dump(per, open(os.path.join(out, f"{tag}_config2_pmc_summary.json"), "w"), indent=1)
    rows = per.get("rows", {})
    fetch_KiB, write_KiB = rows.get("FETCH_SIZE"), rows.get("WRITE_SIZE")
    cal = {}
    if "fill" in per and per["fill"].get("WRITE_SIZE"):
        cal["fill_WRITE_SIZE_bytes_over_known"] = per["fill"]["WRITE_SIZE"] * 1024.0 / (N * 48)
    if "mul" in per:
        if per["mul"].get("WRITE_SIZE"):
            cal["mul_WRITE_SIZE_bytes_over_known"] = per["mul"]["WRITE_SIZE"] * 1024.0 / (N * 48)
        if per["mul"].get("FETCH_SIZE"):
            cal["mul_FETCH_SIZE_bytes_over_known"] = per["mul"]["FETCH_SIZE"] * 1024.0 / (N * 48)   # the guide: 0.5 for wide coalesced reads
    known_written = 65.0   # 64 B row record + the valid byte, per point, unconditional
    if write_KiB:
        cal["rows_WRITE_SIZE_bytes_over_known_65B_per_point"] = write_KiB * 1024.0 / (N * known_written)
    latest = {"tag": tag, "commit": os.environ.get("LK_PROF_COMMIT", "unknown"), "kernel": "lk_residual_kernel<true> (lk_batch_residuals_dev)", "slots": slots,
              "points_per_launch": N, "launches_profiled": launches, "kernel_sources_sha16": bench.kernel_sources_sha16(),
              "FETCH_SIZE_KiB": fetch_KiB, "WRITE_SIZE_KiB": write_KiB, "calibration": cal}
    if fetch_KiB is not None and write_KiB is not None:
        hbm = (2.0 * fetch_KiB + write_KiB) * 1024.0
        latest.update({"hbm_bytes_per_launch": hbm, "hbm_bytes_per_point": hbm / N, "fetch_bytes_per_point_x2": 2.0 * fetch_KiB * 1024.0 / N,
                       "write_bytes_per_point": write_KiB * 1024.0 / N,
                       "hbm_formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 64 B per 128-B request; WRITE_SIZE as read - see calibration)"})
    if rows.get("SQ_WAVES"):
        w = rows["SQ_WAVES"]
        latest.update({"waves": w, "valu_insts_per_wave": rows.get("SQ_INSTS_VALU", 0) / w, "salu_insts_per_wave": rows.get("SQ_INSTS_SALU", 0) / w,
                       "vmem_rd_insts_per_wave": rows.get("SQ_INSTS_VMEM_RD", 0) / w, "vmem_wr_insts_per_wave": rows.get("SQ_INSTS_VMEM_WR", 0) / w,
                       "lds_insts_per_wave": rows.get("SQ_INSTS_LDS", 0) / w})
    if rows.get("SQ_WAVE_CYCLES"):
        latest["wait_any_frac_of_wave_cycles"] = rows.get("SQ_WAIT_ANY", 0) / rows["SQ_WAVE_CYCLES"]
        latest["active_vmem_frac_of_wave_cycles"] = rows.get("SQ_ACTIVE_INST_VMEM", 0) / rows["SQ_WAVE_CYCLES"]
    if rows.get("TCC_REQ_sum"):
        latest["tcc_req_per_point"] = rows["TCC_REQ_sum"] / N
        latest["tcc_hit_rate"] = rows.get("TCC_HIT_sum", 0) / max(1.0, rows.get("TCC_HIT_sum", 0) + rows.get("TCC_MISS_sum", 0))
    if rows.get("avg_dur_us_trace_pass"):
        latest["profiled_dur_us"] = rows["avg_dur_us_trace_pass"]
        if latest.get("hbm_bytes_per_launch"):
            latest["hbm_GBs_at_profiled_dur"] = latest["hbm_bytes_per_launch"] / (rows["avg_dur_us_trace_pass"] * 1e-6) / 1e9
    json.dump(latest, open(os.path.join(out, "latest_config2_pmc.json"), "w"), indent=1)
    print(json.dumps(latest, indent=1))


if __name__ == "__main__":
    main()
