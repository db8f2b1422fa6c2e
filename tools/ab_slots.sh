#!/bin/bash
# residual-kernel time per point vs batch size (grid fill / ramp / tail diagnostic)
for s in ${SLOTS_LIST:-32 64 128 256 512}; do
  LEGKILO_REPLAY_GROUPS=${G:-1} python bench.py --steps 6 --warmup 2 --cpu-sample 0 --stream-scans 0 --scans-per-gpu $s 2>/dev/null > /tmp/ab_s.json
  python - $s <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_s.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("slots", sys.argv[1], "scans/s", d["value"], "ms/step", d["ms_per_step"], "residual_us", round(r["launch_ms_single_stream_events"] * 1e3, 1),
      "ps/pt", round(r["launch_ms_single_stream_events"] * 1e9 / r["points_per_launch"], 2), "other", r["other_kernels_ms"])
PY
done
