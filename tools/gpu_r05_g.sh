#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/gpu_prof_overlay_r05.sh r05c "stats fetch write sq" af88187431dd | tail -n 20
cp gpurun_out/prof_overlay_r05c/latest_overlay_pmc.json profiles/latest_overlay_pmc.json
mkdir -p gpurun_out/r05n
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05n/bench.json 2> gpurun_out/r05n/bench.err; echo "bench rc $?"; tail -c 1500 gpurun_out/r05n/bench.json; tail -n 3 gpurun_out/r05n/bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
