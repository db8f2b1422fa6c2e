#!/bin/bash
# Multi-GPU readiness without the node: bench.py's N = 8 path end to end on the ONE GPU of a gpurun box - eight ranks, gloo instead
# of RCCL (RCCL refuses two ranks on one device), BASELINE config 5 as worded: --total-scans 1024 = 128 scans per rank (strong scaling).  Rank 0 builds
# the map, the device blob goes to the other seven ranks by broadcast AND by scatter + all-gather, every rank replays its block of 128 scans, the
# per-step poses and the per-scan state records are all-gathered, and every rank replays its block WITH insert (overlay) as well.
# NOT a performance figure: eight processes share one GPU and the collectives go through host memory.  Output -> gpurun_out/multirank8_selftest.log
export LEGKILO_BENCH_BACKEND=gloo LEGKILO_BENCH_SHARE_GPU=1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
COMMON="--gpus 8 --steps 3 --warmup 1 --cpu-sample 0 --stream-scans 0 --config1-scans 0 --config4-scans 0 --config2-scans 0 --no-pcie --sustained-s 0 --shuffle-check 0 --gen-workers 8"
{
  echo "# strong scaling, BASELINE config 5 as worded: 1024 scans in total, 128 per rank, 8 ranks on one GPU over gloo";
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29527 \
      bench.py $COMMON --total-scans 1024 --overlay-scans 128 2>&1 | grep -v "libdrm\|OMP_NUM_THREADS\|^\*\*\*\*\|Gloo\]" | tail -30
} > $OUT/multirank8_selftest.log 2>&1
tail -c 2500 $OUT/multirank8_selftest.log
