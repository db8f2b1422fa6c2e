#!/bin/bash
# register budget of the generic root pass of the uniform batch with insert (LEGKILO_OV_ROOT_WAVES = 2 / 3 / 4 waves per SIMD), interleaved on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --config1-scans 0 --no-pcie --sustained-s 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --config4-scans 0 --steps 3 --warmup 1"
$B > /dev/null 2>&1
for rep in 1 2; do
  for g in 2 3 4; do
    LEGKILO_OV_ROOT_WAVES=$g $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; p=d.get('parity_check') or {}
print('OV_ROOT_WAVES=$g rep $rep: overlay', e.get('overlay_ms_per_batch'), 'ms; insert_root', e.get('overlay_kernel_ms_per_batch',{}).get('ov_insert_root'), '; parity', p.get('ok'), (p.get('overlay') or {}).get('counts_equal'))"
  done
done
