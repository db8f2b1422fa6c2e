#!/bin/bash
# Round 5 stream path A/B: library builds x env knobs on the three single-stream shapes (tools/stream_workload.py)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r05s
mkdir -p $OUT
run() {  # label, lib, env..., -- kinds
  local label=$1 lib=$2; shift 2
  for kind in 5 51 vlp; do
    r=$(env LEGKILO_HIP_LIB=$REPO/leg-kilo_amd/$lib "$@" timeout 300 python tools/stream_workload.py --kind $kind --scans 12 --reps 3 2>/dev/null | tail -n 1)
    echo "$label kind=$kind $r" | cut -c1-300 | tee -a $OUT/ab.txt
  done
}
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "update_points or map_update or sequence_imu or config3_full_size or golden" -p no:cacheprovider 2>&1 | tail -n 3
run base liblegkilo_hip.so LK_X=0
run rw3 liblegkilo_rw3.so LK_X=0
run rw3_grid768 liblegkilo_rw3.so LEGKILO_ROOT_GRID=768
run base_grid1024 liblegkilo_hip.so LEGKILO_ROOT_GRID=1024
