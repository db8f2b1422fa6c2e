#!/usr/bin/env python
"""Average PMC values per launch of the batched residual kernel from the passes of tools/gpu_pmc.sh."""
import glob, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01b"
out = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_*"))):
    f = os.path.join(d, "bench_results.db")
    if not os.path.exists(f):
        continue
    c = sqlite3.connect(f).cursor()
    acc = {}
    rows = list(c.execute("select kernel_name, grid_size_y, counter_name, value, duration from counters_collection"))
    gy_max = max([gy for name, gy, _, _, _ in rows if "lk_residual_kernel<false" in name] or [0])   # the full batch only (bench.py also times 128-slot shards)
    for name, gy, cn, val, dur in rows:
        if "lk_residual_kernel<false" in name and gy > 1 and gy == gy_max:
            a = acc.setdefault(cn, [0, 0.0, 0.0])
            a[0] += 1; a[1] += val; a[2] += dur
    for cn, (n, v, du) in acc.items():
        out[cn] = {"launches": n, "avg": v / n, "avg_dur_us_profiled": du / n / 1e3}
json.dump(out, open(os.path.join(os.environ.get("LK_PROFILES_DIR", os.path.join(ROOT, "profiles")), f"{tag}_pmc_attrib.json"), "w"), indent=1)
for k, v in out.items():
    print(f"{k:45s} {v['avg']:16.1f}   (n={v['launches']}, dur {v['avg_dur_us_profiled']:.1f} us)")
