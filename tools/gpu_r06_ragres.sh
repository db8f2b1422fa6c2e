#!/bin/bash
# A/B of the scan-resident recorded-run replay with insert (LEGKILO_RAG_RESIDENT) against the launch-by-launch form, interleaved on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --config4-scans 0 --steps 3 --warmup 1"
$B > /dev/null 2>&1
for rep in 1 2; do
  for m in 0 1; do
    LEGKILO_RAG_VERBOSE=1 LEGKILO_RAG_RESIDENT=$m $B 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; p=d['parity_check']
print('RAG_RESIDENT=$m rep $rep: ragged overlay', e.get('config1_overlay_ragged_ms_per_batch'), 'ms; parity', p['ok'], p.get('config1_overlay_ragged'))"
    grep "scan-resident" /tmp/err.txt | sort | uniq -c | head -3
  done
done
