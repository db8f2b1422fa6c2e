#!/bin/bash
# A/B of the fused tail of a bucket index in the recorded-run batch with insert (LEGKILO_RAG_TAIL, LEGKILO_RAG_TAIL_THREADS) and of the middle kernel's
# workgroup size (LEGKILO_RAG_MID_THREADS), interleaved on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --config4-scans 0 --steps 3 --warmup 1"
$B > /dev/null 2>&1
for rep in 1 2; do
  for m in "0 256 256" "1 64 256" "1 64 128" "1 64 64"; do
    set -- $m
    LEGKILO_RAG_TAIL=$1 LEGKILO_RAG_TAIL_THREADS=$2 LEGKILO_RAG_MID_THREADS=$3 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; p=d['parity_check']
print('RAG_TAIL=$1 tail threads $2 mid threads $3 rep $rep: ragged overlay', e.get('config1_overlay_ragged_ms_per_batch'), 'ms; parity', p['ok'], p.get('config1_overlay_ragged'))"
  done
done
