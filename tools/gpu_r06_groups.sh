#!/bin/bash
# slot groups (HIP streams) of the uniform batch with insert: LEGKILO_OV_GROUPS = 1 .. kMaxGroups, interleaved on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --config1-scans 0 --no-pcie --sustained-s 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --config4-scans 0 --steps 3 --warmup 1"
$B > /dev/null 2>&1
for rep in 1 2; do
  for g in 3 4 6 8; do
    LEGKILO_OV_GROUPS=$g $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; p=d.get('parity_check') or {}
print('OV_GROUPS=$g rep $rep: overlay', e.get('overlay_ms_per_batch'), 'ms; parity', p.get('ok'), (p.get('overlay') or {}).get('counts_equal'))"
  done
done
