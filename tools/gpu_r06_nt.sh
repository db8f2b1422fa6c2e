#!/bin/bash
# A/B of non-temporal row stores in the config-2 rows launch (LK_ROWS_NT), interleaved on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python $REPO/tools/config2_workload.py --cache-dir /tmp/lkcache --slots 256 --reps 2 --no-calib > /dev/null 2>&1
for rep in 1 2 3; do
  for lib in liblegkilo_hip.so liblegkilo_hip_nt0.so; do
    echo -n "$lib: "; LEGKILO_HIP_LIB=$REPO/leg-kilo_amd/$lib python $REPO/tools/config2_workload.py --cache-dir /tmp/lkcache --slots 256 --reps 10 --no-calib 2>/dev/null | tail -1 | cut -c1-140
  done
done
