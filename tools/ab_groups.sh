#!/bin/bash
# A/B of the batch-replay slot-group count (debug knob LEGKILO_REPLAY_GROUPS).
for g in ${GROUPS_LIST:-1 2 3 4}; do
  LEGKILO_REPLAY_GROUPS=$g python bench.py --steps 10 --warmup 2 --cpu-sample 0 --stream-scans 0 2>/dev/null > /tmp/ab_$g.json
  python - $g <<'PY'
import json, sys
g = sys.argv[1]
d = json.loads(open(f"/tmp/ab_{g}.json").read().strip().splitlines()[-1])
print("groups", g, "scans/s", d["value"], "ms/step", d["ms_per_step"])
PY
done
