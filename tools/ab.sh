#!/bin/bash
# A/B the residual kernel variants built as leg-kilo_amd/lib_var_*.so (one bench run each)
for f in leg-kilo_amd/lib_var_*.so; do
  LEGKILO_HIP_LIB=$PWD/$f python bench.py --steps 4 --warmup 1 --cpu-sample 0 --stream-scans 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f', 'value', d['value'], 'res_ms', r['avg_launch_ms'], 'frac', r['frac'], 'other', r['other_kernels_ms'], 'stream_ms', d['extra']['stream_ms_per_scan'])"
done
