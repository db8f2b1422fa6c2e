#!/bin/bash
# Round 5: all-slots parity record, overlay counters, full GPU suite, bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r05k
mkdir -p $OUT
timeout 900 python tools/parity_all_slots.py --out $OUT/r05_parity_all_slots.json > $OUT/parity.log 2>&1; echo "parity_all_slots rc $?"; tail -n 3 $OUT/parity.log | cut -c1-600
bash tools/gpu_prof_overlay_r05.sh r05a "stats fetch write sq" ca074557735c
cp gpurun_out/prof_overlay_r05a/*.json gpurun_out/prof_overlay_r05a/*.csv $OUT/ 2>/dev/null
cp gpurun_out/prof_overlay_r05a/latest_overlay_pmc.json profiles/latest_overlay_pmc.json 2>/dev/null
cd $REPO
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -n 5 $OUT/pytest.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 3000 $OUT/bench.json; tail -n 3 $OUT/bench.err
