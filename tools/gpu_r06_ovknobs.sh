#!/bin/bash
# launch-shape knobs of the uniform batch with insert, one at a time against the defaults, interleaved on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --config1-scans 0 --no-pcie --sustained-s 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --config4-scans 0 --steps 3 --warmup 1"
$B > /dev/null 2>&1
for rep in 1 2; do
  for kv in "LEGKILO_OV_FIT_BLOCKS=12" "LEGKILO_OV_FIT_BLOCKS=16" "LEGKILO_OV_FIT_BLOCKS=24"; do
    env $kv $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; p=d.get('parity_check') or {}; k=e.get('overlay_kernel_ms_per_batch',{})
print('$kv rep $rep: overlay', e.get('overlay_ms_per_batch'), 'ms; fit', k.get('ov_fit_lane'), k.get('ov_fit_eig'), 'mat', k.get('ov_materialise'), 'apply', k.get('ov_insert_apply'), 'root', k.get('ov_insert_root'), '; parity', p.get('ok'), (p.get('overlay') or {}).get('counts_equal'))"
  done
done
