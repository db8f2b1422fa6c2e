#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/gpu_prof_ragov.sh (/tmp/rgp_{stats,fetch,write,sq}) into
  <tag>_ragged_overlay_kernel_stats.csv   the --stats table of the kernel-trace pass (the whole bench process: the frozen-map steps are in it too)
  <tag>_ragged_overlay_pmc.json           per kernel of the recorded-run batch WITH insert and per REPLAY: launches, time, HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, KiB
                                          counters, reads doubled on gfx950), VALU instructions, waves, share of wave cycles spent waiting; per bucket index the same
in $LK_PROFILES_DIR (default profiles/).   usage: collect_ragov_pmc.py <tag> <unprofiled bench line>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("lk_rag_ov_scan_kernel", "lk_rag_ov_front_kernel", "lk_ov_mid_kernel", "lk_ov_insert_root_kernel", "lk_ov_fit_eig_kernel", "lk_ov_fit_group_kernel", "lk_ov_insert_apply_kernel",
           "lk_ov_insert_fallback_kernel", "lk_ov_reset_kernel", "lk_ov_frozen_bits_kernel", "lk_ov_base_sums_kernel", "lk_ov_status_kernel")


def short(name):
    return name.replace("void ", "").split("(")[0].split("<")[0]


def main():
    tag, line = sys.argv[1], sys.argv[2]
    out = os.environ.get("LK_PROFILES_DIR", os.path.join(ROOT, "profiles"))
    os.makedirs(out, exist_ok=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("/tmp/rgp_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    n_disp, t_ns = collections.Counter(), collections.Counter()
    for f in glob.glob("/tmp/rgp_stats/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            n_disp[k] += 1
            t_ns[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for f in glob.glob("/tmp/rgp_stats/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, os.path.join(out, f"{tag}_ragged_overlay_kernel_stats.csv"))
    replays = n_disp.get("lk_ov_reset_kernel", 0)
    if not replays:
        print("no overlay replay in the trace")
        return 1
    indices = n_disp.get("lk_rag_ov_front_kernel", 0) / replays   # 0 in the scan-resident form: the whole chain is lk_rag_ov_scan_kernel
    try:
        e = json.loads(open(line).read().strip().splitlines()[-1])["extra"]
        unprof = {k: v for k, v in e.items() if k.startswith("config1_overlay_ragged")}
    except Exception as ex:   # the profile stands without it
        unprof = {"error": str(ex)}
    kernels, tot_ms, tot_b = {}, 0.0, 0.0
    for k in KERNELS:
        if not n_disp.get(k):
            continue
        c = acc.get(k, {})
        ms = t_ns[k] / 1e6 / replays
        b = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0 / replays
        kernels[k] = {"launches_per_replay": n_disp[k] / replays, "avg_us_per_launch": t_ns[k] / 1e3 / n_disp[k], "ms_per_replay": ms,
                      "hbm_MB_per_replay": b / 1e6, "hbm_GBs_over_its_time": (b / 1e9) / (ms / 1e3) if ms else None,
                      "valu_insts_per_replay": c.get("SQ_INSTS_VALU", 0.0) / replays, "waves_per_replay": c.get("SQ_WAVES", 0.0) / replays,
                      "wait_any_frac_of_wave_cycles": (c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None}
        tot_ms += ms
        tot_b += b
        print(f"{k:32s} {n_disp[k] / replays:7.1f} launches  {t_ns[k] / 1e3 / n_disp[k]:7.1f} us each  {ms:7.2f} ms/replay  HBM {b / 1e6:8.1f} MB/replay")
    sys.path.insert(0, ROOT)
    import bench  # noqa: E402

    res = {"tag": tag, "commit": os.environ.get("LK_PROF_COMMIT", "unknown"), "kernel_sources_sha16": bench.kernel_sources_sha16(bench.RAGOV_KERNEL_SOURCES),
           "what": "tools/gpu_prof_ragov.sh: rocprofv3 kernel trace + PMC passes of bench.py with only the recorded-run batch WITH insert on (1 024 config-1 scans); counters summed "
                   "over all dispatches of a kernel and divided by the replays in the run (= launches of lk_ov_reset_kernel); FETCH_SIZE / WRITE_SIZE in KiB, reads doubled (gfx950)",
           "replays_profiled": replays, "bucket_indices_per_replay": indices, "unprofiled_line": unprof, "kernels": kernels,
           "sum_of_kernel_ms_per_replay": tot_ms, "kernel_us_per_bucket_index": tot_ms * 1e3 / indices if indices else None,
           "hbm_GB_per_replay": tot_b / 1e9, "hbm_frac_of_8TBs_over_kernel_time": (tot_b / 1e9) / (tot_ms / 1e3) / 8000.0 if tot_ms else None}
    json.dump(res, open(os.path.join(out, f"{tag}_ragged_overlay_pmc.json"), "w"), indent=1)
    json.dump(res, open(os.path.join(out, "latest_ragged_overlay_pmc.json"), "w"), indent=1)   # what bench.py's extra.config1_overlay_ragged_roofline reads
    print({k: v for k, v in res.items() if k not in ("kernels", "what")})
    return 0


if __name__ == "__main__":
    sys.exit(main())
