#!/bin/bash
# Round-5 profiling of the batch replay WITH insert (lk_batch_replay_overlay_dev) on the GPU box: one kernel-trace pass and separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass), each its own rocprofv3 run of tools/overlay_workload.py (1024 slots, 32 distinct scans).
# tools/collect_overlay_pmc.py condenses them into <tag>_overlay_kernel_stats.csv, <tag>_overlay_pmc_summary.json and latest_overlay_pmc.json
# (what bench.py's extra.overlay_roofline reads).   usage: tools/gpu_prof_overlay_r05.sh <tag> [passes] [commit]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05a}
PASSES=${2:-"stats fetch write sq"}
export LK_PROF_COMMIT=${3:-unknown}
OUT=$REPO/gpurun_out/prof_overlay_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPS=2
W="python $REPO/tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32 --slots 1024 --reps $REPS --no-profile"
timeout 300 python $REPO/tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32 --slots 64 --reps 1 > $OUT/warm.log 2>&1 < /dev/null   # fills the input cache outside any profiler
for p in $PASSES; do
  rm -rf /tmp/ovp_$p
  case $p in
    stats) timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ovp_stats -o t -- $W > $OUT/stats.log 2>&1 < /dev/null ;;
    fetch) timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/ovp_fetch -o t -- $W > $OUT/fetch.log 2>&1 < /dev/null ;;
    write) timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/ovp_write -o t -- $W > $OUT/write.log 2>&1 < /dev/null ;;
    sq)    timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/ovp_sq -o t -- $W > $OUT/sq.log 2>&1 < /dev/null ;;
  esac
  echo "$p rc=$? $(tail -n 1 $OUT/$p.log | cut -c1-200)"
done
export LK_PROFILES_DIR=$OUT
python $REPO/tools/collect_overlay_pmc.py $TAG $((REPS + 1)) 1024 > $OUT/collect.log 2>&1
tail -n 30 $OUT/collect.log
find $OUT -name '*.log' -size +1M -delete
