#!/bin/bash
# A/B of two builds of the library on the bench headline and the config-2 rows launch, interleaved on one box:  tools/gpu_r06_ab.sh <libA.so> <libB.so>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; Bl=$2
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --config1-scans 0 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0"
$B > /dev/null 2>&1   # fills the cache
for rep in 1 2 3; do
  for lib in $A $Bl; do
    LEGKILO_HIP_LIB=$REPO/leg-kilo_amd/$lib $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib rep $rep: value', d['value'], 'ms/step', d['ms_per_step'], 'parity', d['parity_check']['ok'], d['parity_check']['counts_equal'])"
    echo -n "   config2: "; LEGKILO_HIP_LIB=$REPO/leg-kilo_amd/$lib python $REPO/tools/config2_workload.py --cache-dir /tmp/lkcache --slots 256 --reps 10 --no-calib 2>/dev/null | tail -1 | cut -c30-120
  done
done
