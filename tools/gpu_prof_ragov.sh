#!/bin/bash
# Recorded-run batch WITH insert (lk_batch_replay_overlay_ragged_dev; bench extra config1_overlay_ragged_*: 1 024 config-1 scans, ~370 bucket indices):
# one kernel-trace pass and separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ), each its own rocprofv3 run of bench.py with every other extra off;
# tools/collect_ragov_pmc.py condenses them.   usage: tools/gpu_prof_ragov.sh <tag> [commit]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
export LK_PROF_COMMIT=${2:-unknown}
OUT=$REPO/gpurun_out/prof_ragov_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-sample 0 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --config4-scans 0 --steps 2 --warmup 1 --cache-dir /tmp/lkcache"
timeout 400 python $REPO/bench.py $ARGS > $OUT/warm.json 2>/dev/null < /dev/null   # fills the input cache outside any profiler; its line = the unprofiled time
for p in stats fetch write sq; do
  rm -rf /tmp/rgp_$p
  case $p in
    stats) timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rgp_stats -o t -- python $REPO/bench.py $ARGS > $OUT/stats.json 2> $OUT/stats.log < /dev/null ;;
    fetch) timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rgp_fetch -o t -- python $REPO/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.log < /dev/null ;;
    write) timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/rgp_write -o t -- python $REPO/bench.py $ARGS > $OUT/write.json 2> $OUT/write.log < /dev/null ;;
    sq)    timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/rgp_sq -o t -- python $REPO/bench.py $ARGS > $OUT/sq.json 2> $OUT/sq.log < /dev/null ;;
  esac
  echo "$p rc=$?"
done
export LK_PROFILES_DIR=$OUT
python $REPO/tools/collect_ragov_pmc.py $TAG $OUT/warm.json > $OUT/collect.log 2>&1
tail -n 30 $OUT/collect.log
find $OUT -name '*.log' -size +1M -delete
