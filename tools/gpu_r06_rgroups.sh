#!/bin/bash
# slot groups (HIP streams) of the frozen-map batch step (the headline): LEGKILO_REPLAY_GROUPS = 1 .. 4, interleaved on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --config1-scans 0 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --config4-scans 0"
$B > /dev/null 2>&1
for rep in 1 2 3; do
  for g in 2 3 4; do
    LEGKILO_REPLAY_GROUPS=$g $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity_check') or {}
print('REPLAY_GROUPS=$g rep $rep: value', d['value'], 'ms/step', d['ms_per_step'], 'parity', p.get('ok'), p.get('counts_equal'))"
  done
done
