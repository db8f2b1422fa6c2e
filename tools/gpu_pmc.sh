#!/bin/bash
# PMC attribution of the batched residual kernel (separate passes; SQ has 8 slots, TCP/TA/TCC fewer).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${1:-r01b}
ONLY=${2:-}   # optional space-separated list of pass numbers to run, e.g. "1 2 7"
ARGS="--steps 2 --warmup 0 --cpu-sample 0 --stream-scans 0 --map-warm 4 --unique-scans 8"
cd /tmp && export TMPDIR=/tmp
export LEGKILO_REPLAY_GROUPS=1   # whole-batch launches, comparable across rounds
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $i "; then continue; fi
  timeout 240 rocprofv3 --pmc $line -d $OUT/pmc_${TAG}_$i -o bench -- python $REPO/bench.py $ARGS > $OUT/pmc_${TAG}_$i.log 2>&1 < /dev/null
  echo "pass $i: $line -> $(ls $OUT/pmc_${TAG}_$i 2>/dev/null | head -1) $(grep -ciE 'error|invalid' $OUT/pmc_${TAG}_$i.log)"
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAVES
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
PASSES
find $OUT -name '*.db' -size +30M -delete
