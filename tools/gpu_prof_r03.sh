#!/bin/bash
# Round-3 profiling of the bench command on the GPU box (via gpurun): one kernel-trace pass and five PMC passes, each its own
# rocprofv3 run (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass; SQ has 8 slots).  Every pass runs bench.py with ALL 1024
# distinct scans (generated once, cached under /tmp for the later passes).  tools/collect_r03.py condenses the outputs into
# profiles/<tag>_kernel_stats.csv, <tag>_pmc_residual.json, <tag>_pmc_attrib.json and profiles/latest_pmc.json (what bench.py reads).
#   usage: tools/gpu_prof_r03.sh <tag> [passes] [commit]   passes: subset of "stats fetch write sq1 sq2 sq3 tcc" (default all); sq3 = the fp64 instruction counters
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${1:-r03a}
PASSES=${2:-"stats fetch write sq1 sq2 sq3 tcc"}
export LK_PROF_COMMIT=${3:-unknown}   # commit the snapshot was taken from (the GPU box has no .git)
COMMON=${COMMON:-"--cpu-sample 0 --config1-scans 0 --no-pcie --sustained-s 0 --cache-dir /tmp/lkcache"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, extra bench args, rocprof args...
  local name=$1 bargs=$2; shift 2
  timeout 420 rocprofv3 "$@" -d $OUT/$name -o bench -- python $REPO/bench.py $COMMON $bargs > $OUT/$name.log 2>&1 < /dev/null
  echo "$name: rc=$? $(ls $OUT/$name 2>/dev/null | head -1) errors=$(grep -ciE 'error|invalid' $OUT/$name.log)"
}
# populate the input cache OUTSIDE rocprofv3: the generator forks a worker pool, and forking a process that has the profiler's
# counter tool (HSA already initialised) loaded can hang
timeout 300 python $REPO/bench.py $COMMON --steps 1 --warmup 0 --stream-scans 3 > $OUT/prof_${TAG}_warm.log 2>&1 < /dev/null
for p in $PASSES; do
  case $p in
    stats) run prof_${TAG}_stats "--steps 3 --warmup 1 --stream-scans 3" --kernel-trace --stats ;;
    fetch) run prof_${TAG}_fetch "--steps 2 --warmup 0 --stream-scans 0" --pmc FETCH_SIZE ;;
    write) run prof_${TAG}_write "--steps 2 --warmup 0 --stream-scans 0" --pmc WRITE_SIZE ;;
    sq1)   run pmc_${TAG}_1 "--steps 2 --warmup 0 --stream-scans 0" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS ;;
    sq2)   run pmc_${TAG}_2 "--steps 2 --warmup 0 --stream-scans 0" --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE ;;
    sq3)   run pmc_${TAG}_4 "--steps 2 --warmup 0 --stream-scans 0" --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_WAVES ;;
    tcc)   run pmc_${TAG}_3 "--steps 2 --warmup 0 --stream-scans 0" --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ;;
  esac
done
# condense ON the box (the rocpd databases are tens of MB each; gpurun_out/ merges back only below 64 MiB), keep the summaries
export LK_PROFILES_DIR=$OUT/profiles_$TAG
mkdir -p $LK_PROFILES_DIR
python $REPO/tools/collect_r03.py $TAG > $OUT/collect_$TAG.log 2>&1
tail -n 40 $OUT/collect_$TAG.log
rm -rf $OUT/prof_${TAG}_* $OUT/pmc_${TAG}_[0-9]
find $OUT -name '*.log' -size +2M -delete
du -sh $OUT 2>/dev/null
