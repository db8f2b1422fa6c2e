#!/bin/bash
TAG=${1:-r04f}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== round-3 tree, readlane gather: scheduling barrier only / s_nop 1 / 16 nops without a memory clobber"
for v in r3_readlane_barrier r3_readlane_nop1 r3_readlane_nop_nomem r3_readlane r3_shfl; do
  for rep in 1 2; do
  LEGKILO_HIP_LIB=$REPO/tools/probes/liblegkilo_$v.so timeout 300 python -m pytest $REPO/tests/test_gpu_parity.py -q -m gpu -k "test_map_update_surface" -p no:cacheprovider > $OUT/$v.$rep.log 2>&1; echo "$v run $rep rc $? $(grep -o "'npts', [0-9-]*, [0-9-]*" $OUT/$v.$rep.log | head -1)"
  done
done
echo "== bench.py --gpus 2 without a launcher (two ranks sharing the GPU, gloo hook)"
B="--steps 3 --warmup 1 --scans-per-gpu 64 --unique-scans 16 --cpu-sample 0 --config1-scans 0 --stream-scans 0 --no-pcie --sustained-s 0 --overlay-scans 0 --map-warm 4"
LEGKILO_BENCH_SHARE_GPU=1 LEGKILO_BENCH_BACKEND=gloo timeout 600 python $REPO/bench.py --gpus 2 $B > $OUT/selflaunch_gloo.json 2> $OUT/selflaunch_gloo.err; echo "gloo rc $?"; tail -c 400 $OUT/selflaunch_gloo.json
echo "== the same over RCCL (two ranks on ONE device)"
LEGKILO_BENCH_SHARE_GPU=1 timeout 300 python $REPO/bench.py --gpus 2 $B > $OUT/selflaunch_rccl_one_gpu.json 2> $OUT/selflaunch_rccl_one_gpu.err; echo "rccl-on-one-gpu rc $?"; tail -c 300 $OUT/selflaunch_rccl_one_gpu.json; grep -iE "duplicate|invalid usage|ncclInvalid|Error" $OUT/selflaunch_rccl_one_gpu.err | head -5 | cut -c1-300
