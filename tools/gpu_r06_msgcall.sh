#!/bin/bash
# A/B of two builds of the library on the live-stream extras of the bench line (config-1 live, config 4, 5 / 51-bucket streams), interleaved on one box:
#   tools/gpu_r06_msgcall.sh <libA.so> <libB.so>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; Bl=$2
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --config2-scans 0 --steps 3 --warmup 1"
$B > /dev/null 2>&1   # fills the cache
for rep in 1 2 3; do
  for lib in $A $Bl; do
    LEGKILO_HIP_LIB=$REPO/leg-kilo_amd/$lib $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; p=d.get('parity_check') or {}
print('$lib rep $rep: config-1 live', e.get('config1_live_stream_ms_per_scan'), ' config 4 path', e.get('config4_path_ms_per_scan'), ' stream 5 / 51', e.get('stream_ms_per_scan'), e.get('stream51_ms_per_scan'), ' parity', p.get('ok'), p.get('config4_live'))"
  done
done
