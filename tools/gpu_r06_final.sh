#!/bin/bash
# End-of-round evidence at one commit, one box: GPU suite, the suite on poisoned pools, every slot of the bench batch against the oracle, counters of the
# batch with insert (uniform and recorded-run), both bench protocols.   usage: tools/gpu_r06_final.sh <commit> [tag]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
C=${1:-unknown}
TAG=${2:-r06s}
OUT=$REPO/gpurun_out/r06_final
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite: $(tail -n 1 $OUT/suite.log)"
LEGKILO_POISON_POOLS=1 timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > $OUT/suite_poisoned.log 2>&1; echo "poisoned: $(tail -n 1 $OUT/suite_poisoned.log)"
LK_PROF_COMMIT=$C timeout 1500 python tools/parity_all_slots.py --out $OUT/r06_parity_all_slots.json > $OUT/parity.log 2>&1; echo "parity_all_slots rc $?"; tail -n 2 $OUT/parity.log | cut -c1-500
bash tools/gpu_prof_overlay_r05.sh $TAG 'stats fetch write sq' $C > $OUT/ovprof.log 2>&1; tail -n 2 $OUT/ovprof.log
bash tools/gpu_prof_ragov.sh $TAG $C > $OUT/ragov.log 2>&1; tail -n 2 $OUT/ragov.log | cut -c1-400
cp $REPO/gpurun_out/prof_overlay_$TAG/latest_overlay_pmc.json $REPO/profiles/latest_overlay_pmc.json                   # the bench lines below read them
cp $REPO/gpurun_out/prof_ragov_$TAG/latest_ragged_overlay_pmc.json $REPO/profiles/latest_ragged_overlay_pmc.json
cd $REPO
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "driver line rc $?"
timeout 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default line rc $?"
