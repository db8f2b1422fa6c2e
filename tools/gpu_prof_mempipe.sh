#!/bin/bash
# Round 6, VERDICT item 6: WHERE does the graded kernel's s_waitcnt share go?  Memory-pipe counters of lk_residual_kernel<false, 1, ...> in the bench
# command - L1 (TCP) stalls, request latency and its address / data hand-offs with the vector-memory unit, LDS, the SQ's in-flight levels - each group its own rocprofv3
# pass, once on the headline's cell order and once on the batch with a random order inside every bucket (--shuffle-main).
# tools/collect_mempipe.py condenses them into <tag>_pmc_memory_pipe.json.     usage: tools/gpu_prof_mempipe.sh <tag> [commit] ["orders"]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
export LK_PROF_COMMIT=${2:-unknown}
ORDERS=${3:-"cell shuffled"}
OUT=$REPO/gpurun_out/prof_mempipe_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--cpu-sample 0 --config1-scans 0 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --cache-dir /tmp/lkcache --steps 2 --warmup 0"
timeout 300 python $REPO/bench.py $COMMON > $OUT/warm.log 2>&1 < /dev/null
pass() {  # name, order, counters...
  local name=$1 order=$2; shift 2
  local extra=""; [ "$order" = "shuffled" ] && extra="--shuffle-main"
  rm -rf /tmp/mp_${order}_$name
  timeout 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/mp_${order}_$name -o t -- python $REPO/bench.py $COMMON $extra > $OUT/${order}_$name.log 2>&1 < /dev/null
  echo "$order $name rc=$? $(grep -ciE 'error|invalid' $OUT/${order}_$name.log) error lines"
}
for o in $ORDERS; do
  # TA_* and TD_* counters (TA_TA_BUSY, TA_ADDR_STALLED_BY_TC_CYCLES, TA_TOTAL_WAVEFRONTS, TD_TD_BUSY, TD_TC_STALL, ...) are NOT collected: with any of them
  # in --pmc the profiled process never finishes its first dispatch on this pool ("There are 1 incomplete dispatches", killed by the 400 s
  # timeout - six passes of six, round 6).  What the address / data-return path does is read from the L1's side instead (TCP_* below).
  pass tcp1 $o GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
  pass tcp2 $o TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
  pass tcp3 $o TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum
  pass tcp4 $o TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum
  pass sq   $o SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES
  pass sq2  $o SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES
done
export LK_PROFILES_DIR=$OUT
python $REPO/tools/collect_mempipe.py $TAG > $OUT/collect.log 2>&1
tail -n 80 $OUT/collect.log
find $OUT -name '*.log' -size +1M -delete
