#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/gpu_prof_r05.sh r05b cd86a1142844 > gpurun_out/prof_r05b.log 2>&1; tail -n 25 gpurun_out/prof_r05b.log | cut -c1-200
cp gpurun_out/latest_pmc_r05b.json profiles/latest_pmc.json 2>/dev/null
cp gpurun_out/latest_shuffled_pmc_r05b.json profiles/latest_shuffled_pmc.json 2>/dev/null
mkdir -p gpurun_out/r05m
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05m/bench.json 2> gpurun_out/r05m/bench.err; echo "bench rc $?"; tail -c 2500 gpurun_out/r05m/bench.json; tail -n 3 gpurun_out/r05m/bench.err
