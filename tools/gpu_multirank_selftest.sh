#!/bin/bash
# 2 ranks on the one GPU of a gpurun box, gloo instead of RCCL: exercises bench.py's N>1 path end to end
# (map build on rank 0, device-blob export, broadcast AND scatter + all-gather, import on rank 1, sharded replay, pose
# all-gather) in the weak mode and in the strong mode (--total-scans).  Output -> gpurun_out/multirank_selftest.log
# (copied to profiles/ as evidence).  RCCL itself needs one GPU per rank: that run is the driver's.
export LEGKILO_BENCH_BACKEND=gloo LEGKILO_BENCH_SHARE_GPU=1
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $OUT
COMMON="--gpus 2 --steps 3 --warmup 1 --cpu-sample 0 --stream-scans 0 --config1-scans 0 --no-pcie --sustained-s 0 --map-warm 4 --gen-workers 16"
{
  echo "# weak: 32 scans per rank";
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py $COMMON --scans-per-gpu 32 2>&1 | tail -3
  echo "# strong: 64 scans in total";
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 \
      bench.py $COMMON --total-scans 64 2>&1 | tail -3
} > $OUT/multirank_selftest.log 2>&1
tail -c 1500 $OUT/multirank_selftest.log
