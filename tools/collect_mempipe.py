#!/usr/bin/env python3
"""Condenses the passes of tools/gpu_prof_mempipe.sh (/tmp/mp_<order>_<pass>) into <tag>_pmc_memory_pipe.json in $LK_PROFILES_DIR (default profiles/):
per order (cell / shuffled) the average per DISPATCH of the batch residual kernel (lk_residual_kernel<false, 1, ...>) of every counter, the kernel's
average duration in the same pass, and a few ratios that answer "where does the wave wait":
  * per-CU busy fractions of the vector-memory address unit (TA) and the data-return unit (TD): *_BUSY_sum / (n_units x GRBM_GUI_ACTIVE)
  * L1 (TCP): average latency of a read request to L2 in cycles (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ), average L1 residency of an access
    (TCP_TCP_LATENCY / TCP_TOTAL_ACCESSES), stall cycles by cause per access
  * LDS: bank-conflict cycles per LDS-active cycle, instructions per wave
  * SQ: average number of vector-memory / LDS instructions in flight per wave (SQ_INST_LEVEL_* / SQ_WAVE_CYCLES)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N_CU = 256
FULL_GRID = 313 * 64 * 1024   # threads of one batch residual launch: 313 tiles x 1024 scans


def main():
    tag = sys.argv[1]
    out = os.environ.get("LK_PROFILES_DIR", os.path.join(ROOT, "profiles"))
    res = {"tag": tag, "commit": os.environ.get("LK_PROF_COMMIT", "unknown"), "kernel": "lk_residual_kernel<false, 1, XID> (batch replay, frozen-map grid)",
           "kernel_sources_sha16": bench.kernel_sources_sha16(),
           "what": "tools/gpu_prof_mempipe.sh: rocprofv3 --pmc passes of `bench.py --steps 2` (1024 x 20 000 points per launch); averages per dispatch"}
    for order in ("cell", "shuffled"):
        acc = collections.defaultdict(float)
        cnt = collections.defaultdict(set)
        durs = []
        for d in glob.glob(f"/tmp/mp_{order}_*"):
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if "lk_residual_kernel<false, 1" not in k and "lk_residual_kernel<(bool)0, 1" not in k:
                        continue
                    if int(float(r.get("Grid_Size") or 0)) != FULL_GRID:   # the 1024-scan launches of the timed loop only (the shard128 extra launches the same kernel over 128 scans)
                        continue
                    acc[r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
            for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    gx, gy = int(float(r.get("Grid_Size_X") or 0)), int(float(r.get("Grid_Size_Y") or 0))   # the kernel trace lists the grid per dimension
                    full = (gx * max(gy, 1) == FULL_GRID) if gx else (int(float(r.get("Grid_Size") or 0)) == FULL_GRID)
                    if ("lk_residual_kernel<false, 1" in k or "lk_residual_kernel<(bool)0, 1" in k) and full:
                        durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
        if not acc:
            continue
        c = {k: v / max(1, len(cnt[k])) for k, v in acc.items()}
        g = c.get
        o = {"counters_avg_per_dispatch": c, "dispatches_counted": {k: len(v) for k, v in cnt.items()}, "avg_dur_us_under_profiler": sum(durs) / max(1, len(durs))}
        gui = g("GRBM_GUI_ACTIVE")
        pts = 1024 * 20000
        waves = g("SQ_WAVES") or (pts / 64.0)
        d = {}
        if gui:
            if g("TA_TA_BUSY_sum") is not None:
                d["ta_busy_frac_per_cu"] = g("TA_TA_BUSY_sum") / (N_CU * gui)
            if g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") is not None:
                d["ta_addr_stalled_by_tc_frac_per_cu"] = g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") / (N_CU * gui)
            if g("TA_DATA_STALLED_BY_TC_CYCLES_sum") is not None:
                d["ta_data_stalled_by_tc_frac_per_cu"] = g("TA_DATA_STALLED_BY_TC_CYCLES_sum") / (N_CU * gui)
        if g("TA_BUSY_avr") is not None:
            d["ta_busy_avr_percent_as_reported"] = g("TA_BUSY_avr")
        if g("TCP_TCC_READ_REQ_sum"):
            d["tcp_read_req_to_l2_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / g("TCP_TCC_READ_REQ_sum")
            d["tcp_read_req_to_l2_per_point"] = g("TCP_TCC_READ_REQ_sum") / pts
            if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
                d["tcp_l1_hit_rate"] = 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
                d["tcp_pending_stall_cycles_per_cache_access"] = g("TCP_PENDING_STALL_CYCLES_sum", 0.0) / g("TCP_TOTAL_CACHE_ACCESSES_sum")
        if g("TCP_TOTAL_ACCESSES_sum"):
            d["tcp_latency_cycles_per_access"] = g("TCP_TCP_LATENCY_sum", 0.0) / g("TCP_TOTAL_ACCESSES_sum")
            d["tcp_accesses_per_point"] = g("TCP_TOTAL_ACCESSES_sum") / pts
            d["tcp_ta_data_stall_cycles_per_access"] = g("TCP_TCP_TA_DATA_STALL_CYCLES_sum", 0.0) / g("TCP_TOTAL_ACCESSES_sum")
            d["tcp_read_tagconflict_stall_cycles_per_access"] = g("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0.0) / g("TCP_TOTAL_ACCESSES_sum")
        if g("TCP_TOTAL_ACCESSES_sum") and g("TCP_TCP_TA_ADDR_STALL_CYCLES_sum") is not None:
            pass
        for k in ("TCP_TCP_TA_ADDR_STALL_CYCLES_sum", "TCP_LFIFO_STALL_CYCLES_sum", "TCP_RFIFO_STALL_CYCLES_sum", "TCP_TCR_RDRET_STALL_sum"):
            if g(k) is not None and g("GRBM_GUI_ACTIVE"):
                d[k.lower().replace("_sum", "") + "_frac_per_cu"] = g(k) / (N_CU * g("GRBM_GUI_ACTIVE"))
        if g("TCP_GATE_EN1_sum"):
            d["tcp_busy_frac_of_clocked"] = g("TCP_GATE_EN2_sum", 0.0) / g("TCP_GATE_EN1_sum")
            d["tcp_tcr_stall_frac_of_clocked"] = g("TCP_TCR_TCP_STALL_CYCLES_sum", 0.0) / g("TCP_GATE_EN1_sum")
        if g("TCP_UTCL1_TRANSLATION_MISS_sum") is not None:
            d["utcl1_translation_misses_per_point"] = g("TCP_UTCL1_TRANSLATION_MISS_sum") / pts
        if g("TD_TD_BUSY_sum") is not None and g("TD_LOAD_WAVEFRONT_sum"):
            d["td_busy_cycles_per_load_wavefront"] = g("TD_TD_BUSY_sum") / g("TD_LOAD_WAVEFRONT_sum")
            d["td_tc_stall_cycles_per_load_wavefront"] = g("TD_TC_STALL_sum", 0.0) / g("TD_LOAD_WAVEFRONT_sum")
            d["td_load_wavefronts_per_wave"] = g("TD_LOAD_WAVEFRONT_sum") / waves
        if g("TA_TOTAL_WAVEFRONTS_sum"):
            d["ta_wavefront_instructions_per_wave"] = g("TA_TOTAL_WAVEFRONTS_sum") / waves
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
                if g(k) is not None:
                    d[k.lower() + "_frac_of_wave_cycles"] = g(k) / wc
            if g("SQ_INST_LEVEL_VMEM") is not None:
                d["vmem_insts_in_flight_per_wave_avg"] = g("SQ_INST_LEVEL_VMEM") / wc
            if g("SQ_INST_LEVEL_LDS") is not None:
                d["lds_insts_in_flight_per_wave_avg"] = g("SQ_INST_LEVEL_LDS") / wc
        if g("SQ_INSTS_LDS"):
            d["lds_insts_per_wave"] = g("SQ_INSTS_LDS") / waves
            if g("SQ_LDS_IDX_ACTIVE"):
                d["lds_bank_conflict_cycles_per_lds_active_cycle"] = g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE")
                d["lds_addr_conflict_cycles_per_lds_active_cycle"] = g("SQ_LDS_ADDR_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE")
        o["derived"] = d
        res[order] = o
    json.dump(res, open(os.path.join(out, f"{tag}_pmc_memory_pipe.json"), "w"), indent=1)
    for order in ("cell", "shuffled"):
        if order in res:
            print(order, "avg dur us", round(res[order]["avg_dur_us_under_profiler"], 1))
            print(json.dumps(res[order]["derived"], indent=1))


if __name__ == "__main__":
    main()
