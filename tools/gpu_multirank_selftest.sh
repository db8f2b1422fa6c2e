#!/bin/bash
# 2 ranks on the one GPU of a gpurun box, gloo instead of RCCL: exercises bench.py's N>1 path end to end
# (map build on rank 0, device-blob export, collective, import on rank 1, sharded replay, result all-gather).
export LEGKILO_BENCH_BACKEND=gloo LEGKILO_BENCH_SHARE_GPU=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 3 --warmup 1 --scans-per-gpu 32 --unique-scans 8 --cpu-sample 0 --stream-scans 0 2>&1 | tail -5
