#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/gpu_prof_overlay_r05.sh (/tmp/ovp_{stats,fetch,write,sq}) into
  <tag>_overlay_kernel_stats.csv   the --stats table of the kernel-trace pass
  <tag>_overlay_pmc_summary.json   per lk_* kernel: counters summed over all dispatches, dispatch count
  latest_overlay_pmc.json          what bench.py's extra.overlay_roofline reads: per overlay kernel and REPLAY the HBM bytes
                                   (2 x FETCH_SIZE + WRITE_SIZE, KiB counters; the guide's gfx950 correction for reads), VALU instructions,
                                   VGPRs / scratch / waves per SIMD, beside the fingerprint of the kernel sources
in $LK_PROFILES_DIR (default profiles/).   usage: collect_overlay_pmc.py <tag> <replays in the profiled run> <slots>"""
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

KEYS = {"ov_reset": "lk_ov_reset_kernel", "ov_residual": "lk_ov_residual_kernel", "ov_begin": "lk_ov_begin_kernel", "ov_reproject": "lk_ov_reproject_kernel",
        "ov_materialise": "lk_ov_materialise_kernel", "ov_point_geom": "lk_ov_point_geom_kernel", "ov_root_lane": "lk_ov_root_lane_kernel",
        "ov_insert_root": "lk_ov_insert_root_kernel", "ov_fit_eig": "lk_ov_fit_eig_kernel", "ov_fit_lane": "lk_ov_fit_group_kernel",   # round 6: the fit pass by groups of eight lanes (bench.py keeps the profile key "ov_fit_lane")
        "ov_insert_apply": "lk_ov_insert_apply_kernel", "ov_insert_fallback": "lk_ov_insert_fallback_kernel", "ov_base_sums": "lk_ov_base_sums_kernel"}


def short(name):
    return name.replace("void ", "").split("(")[0].split("<")[0]


def main():
    tag, replays, slots = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out = os.environ.get("LK_PROFILES_DIR", os.path.join(ROOT, "profiles"))
    os.makedirs(out, exist_ok=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    regs = {}
    for f in glob.glob("/tmp/ovp_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add((f, r["Dispatch_Id"]))
            regs[k] = {"vgprs": int(r.get("VGPR_Count") or 0), "agprs": int(r.get("Accum_VGPR_Count") or 0), "scratch": int(r.get("Scratch_Size") or r.get("Private_Segment_Size") or 0),
                       "lds": int(r.get("LDS_Block_Size") or 0)}
    n_disp = {}
    for f in glob.glob("/tmp/ovp_stats/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            n_disp[k] = n_disp.get(k, 0) + 1
            if k not in regs:
                regs[k] = {"vgprs": int(r.get("VGPR_Count") or 0), "agprs": int(r.get("Accum_VGPR_Count") or 0), "scratch": int(r.get("Scratch_Size") or r.get("Private_Segment_Size") or 0),
                           "lds": int(r.get("LDS_Block_Size") or 0)}
    for f in glob.glob("/tmp/ovp_stats/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, os.path.join(out, f"{tag}_overlay_kernel_stats.csv"))
    summary = {}
    for k, c in sorted(acc.items()):
        if k.startswith("lk_"):
            summary[k] = dict(c)
            summary[k]["dispatches_in_trace_pass"] = n_disp.get(k)
            summary[k].update(regs.get(k, {}))
    json.dump(summary, open(os.path.join(out, f"{tag}_overlay_pmc_summary.json"), "w"), indent=1)
    # registers / scratch / waves per SIMD as the COMPILER reports them (profiles/r05_resource_usage.txt, `make resource-usage`): the profiler's
    # VGPR_Count column is an allocation granule count on gfx950, not the kernel's VGPRs
    table = {}
    rt = os.path.join(ROOT, "profiles", "r06_resource_usage.txt")
    if not os.path.exists(rt):
        rt = os.path.join(ROOT, "profiles", "r05_resource_usage.txt")
    if os.path.exists(rt):
        for line in open(rt).read().splitlines()[1:]:
            f = line.split()
            if len(f) >= 6:
                name = " ".join(f[:-5])
                table.setdefault(name.split("<")[0], []).append((name, int(f[-5]), int(f[-4]), int(f[-3]), int(f[-2])))
    kernels = {}
    for key, kn in KEYS.items():
        c = acc.get(kn)
        if not c:
            continue
        rg = dict(regs.get(kn, {}))
        v = rg.get("vgprs", 0) + rg.get("agprs", 0)
        waves = 8 if v <= 64 else max(1, min(8, 512 // (((v + 7) // 8) * 8)))
        if kn in table:   # the instantiation the replay launches: <true> (LEAN / XID) or <3, true> (generic root pass)
            cand = [t for t in table[kn] if "<" not in t[0] or "true>" in t[0]] or table[kn]
            rg.update({"vgprs": cand[0][1], "agprs": cand[0][2], "scratch": cand[0][3]})
            waves = cand[0][4]
        e = {"hbm_bytes_per_replay": (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0 / replays,
             "fetch_KiB_per_replay": c.get("FETCH_SIZE", 0.0) / replays, "write_KiB_per_replay": c.get("WRITE_SIZE", 0.0) / replays,
             "valu_insts_per_replay": c.get("SQ_INSTS_VALU", 0.0) / replays, "salu_insts_per_replay": c.get("SQ_INSTS_SALU", 0.0) / replays,
             "vmem_rd_insts_per_replay": c.get("SQ_INSTS_VMEM_RD", 0.0) / replays, "vmem_wr_insts_per_replay": c.get("SQ_INSTS_VMEM_WR", 0.0) / replays,
             "waves_per_replay": c.get("SQ_WAVES", 0.0) / replays,
             "wait_any_frac_of_wave_cycles": (c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
             "dispatches_per_replay": (n_disp.get(kn, 0) / replays) if n_disp.get(kn) else None,
             "vgprs": rg.get("vgprs"), "agprs": rg.get("agprs"), "scratch": rg.get("scratch"), "waves_per_simd": waves}
        kernels[key] = e
        print(f"{kn:32s} HBM {e['hbm_bytes_per_replay'] / 1e9:7.2f} GB/replay  VALU {e['valu_insts_per_replay'] / 1e6:8.1f} M  waves {e['waves_per_replay']:10.0f}  "
              f"VGPR {rg.get('vgprs')} scratch {rg.get('scratch')} -> {waves} waves/SIMD  wait_any {e['wait_any_frac_of_wave_cycles']}")
    latest = {"tag": tag, "commit": os.environ.get("LK_PROF_COMMIT", "unknown"), "slots": slots, "replays_profiled": replays,
              "kernel_sources_sha16": bench.kernel_sources_sha16(bench.OV_KERNEL_SOURCES),
              "what": "tools/gpu_prof_overlay_r05.sh: rocprofv3 --pmc passes of tools/overlay_workload.py (1024 slots, 32 distinct scans tiled), counters summed over all dispatches of a kernel "
                      "and divided by the replays in the run; FETCH_SIZE / WRITE_SIZE in KiB, reads doubled (MI355X_MICROARCH.md, gfx950)",
              "kernels": kernels, "total_hbm_GB_per_replay": sum(e["hbm_bytes_per_replay"] for e in kernels.values()) / 1e9}
    json.dump(latest, open(os.path.join(out, "latest_overlay_pmc.json"), "w"), indent=1)
    print("total HBM GB per replay", round(latest["total_hbm_GB_per_replay"], 2))


if __name__ == "__main__":
    main()
