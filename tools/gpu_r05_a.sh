#!/bin/bash
# Round 5, first GPU call: GPU test suite, overlay A/B (fast root pass on / off) with per-kernel times, the bench line.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r05a}
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -n 15 $OUT/pytest.log | cut -c1-300
W="python $REPO/tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32"
timeout 300 $W --slots 64 --reps 1 > $OUT/warm.log 2>&1 < /dev/null
for f in 1 0; do
  echo "LEGKILO_OV_FAST=$f"
  LEGKILO_OV_FAST=$f timeout 300 $W --slots 1024 --reps 3 2>$OUT/ov_fast$f.err | tail -n 1 | tee $OUT/ov_fast$f.json
done
for w in 3 5; do
  echo "LEGKILO_OV_FAST_WAVES=$w"
  LEGKILO_OV_FAST_WAVES=$w timeout 300 $W --slots 1024 --reps 3 2>/dev/null | tail -n 1 | tee $OUT/ov_fastwaves$w.json
done
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 6000 $OUT/bench.json; tail -n 5 $OUT/bench.err
