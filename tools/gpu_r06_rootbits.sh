#!/bin/bash
# A/B of the re-projection pass's "base voxel takes the point at its root" bit (LEGKILO_OV_ROOT_BITS) on the two batch-with-insert extras, interleaved on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --no-pcie --sustained-s 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --config4-scans 0 --steps 3 --warmup 1"
$B > /dev/null 2>&1
for rep in 1 2; do
  for m in 0 1; do
    LEGKILO_OV_ROOT_BITS=$m $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; p=d['parity_check']
print('ROOT_BITS=$m rep $rep: overlay', e.get('overlay_ms_per_batch'), 'ms; reproject', e.get('overlay_kernel_ms_per_batch',{}).get('ov_reproject'), '; ragged overlay', e.get('config1_overlay_ragged_ms_per_batch'), 'ms; parity', p['ok'], p.get('overlay',{}).get('counts_equal'), p.get('overlay',{}).get('max_pos_delta_m'), p.get('config1_overlay_ragged'))"
  done
done
