#!/bin/bash
# Profiling of BASELINE config 2 at bandwidth size (lk_batch_residuals_dev: residual rows of 256 x 100 000 points materialised in HBM) on the GPU
# box: one kernel-trace pass and separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass), each its own rocprofv3 run of
# tools/config2_workload.py.  tools/collect_config2_pmc.py condenses them into <tag>_config2_kernel_stats.csv, <tag>_config2_pmc_summary.json and
# latest_config2_pmc.json (what bench.py's extra.config2_roofline reads).   usage: tools/gpu_prof_config2.sh <tag> [passes] [commit] [slots]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06a}
PASSES=${2:-"stats fetch write sq1 sq2 tcc"}
export LK_PROF_COMMIT=${3:-unknown}
SLOTS=${4:-256}
OUT=$REPO/gpurun_out/prof_config2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPS=4
W="python $REPO/tools/config2_workload.py --cache-dir /tmp/lkcache --slots $SLOTS --reps $REPS"
timeout 400 $W > $OUT/warm.log 2>&1 < /dev/null   # fills the input cache outside any profiler; its JSON line is the unprofiled timing
tail -n 1 $OUT/warm.log | cut -c1-400
for p in $PASSES; do
  rm -rf /tmp/c2p_$p
  case $p in
    stats) timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2p_stats -o t -- $W > $OUT/stats.log 2>&1 < /dev/null ;;
    fetch) timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/c2p_fetch -o t -- $W > $OUT/fetch.log 2>&1 < /dev/null ;;
    write) timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/c2p_write -o t -- $W > $OUT/write.log 2>&1 < /dev/null ;;
    sq1)   timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/c2p_sq1 -o t -- $W > $OUT/sq1.log 2>&1 < /dev/null ;;
    sq2)   timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/c2p_sq2 -o t -- $W > $OUT/sq2.log 2>&1 < /dev/null ;;
    tcc)   timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d /tmp/c2p_tcc -o t -- $W > $OUT/tcc.log 2>&1 < /dev/null ;;
  esac
  echo "$p rc=$? $(tail -n 1 $OUT/$p.log | cut -c1-200)"
done
export LK_PROFILES_DIR=$OUT
python $REPO/tools/collect_config2_pmc.py $TAG $((REPS + 1)) $SLOTS > $OUT/collect.log 2>&1
tail -n 40 $OUT/collect.log
find $OUT -name '*.log' -size +1M -delete
