#!/bin/bash
# full GPU suite + bench line + smoke at HEAD
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/r05m
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5 | tee gpurun_out/r05m/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05m/bench.json 2> gpurun_out/r05m/bench.err; echo "bench rc $?"; tail -n 3 gpurun_out/r05m/bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
