#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench command, then separate
# PMC passes (FETCH_SIZE / WRITE_SIZE need their own runs: TCC has 4 slots, FETCH_SIZE costs 3).
# Outputs land in gpurun_out/prof_*; tools/collect_profiles.py condenses them into profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${1:-r01}
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --stream-scans 3"
T="timeout 240"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_stats -o bench -- python $REPO/bench.py $ARGS > $OUT/prof_${TAG}_stats.log 2>&1
$T rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_${TAG}_fetch -o bench -- python $REPO/bench.py $ARGS > $OUT/prof_${TAG}_fetch.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_${TAG}_write -o bench -- python $REPO/bench.py $ARGS > $OUT/prof_${TAG}_write.log 2>&1
$T rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -d $OUT/prof_${TAG}_sq -o bench -- python $REPO/bench.py $ARGS > $OUT/prof_${TAG}_sq.log 2>&1
# keep only the small CSVs (<= 64 MiB merges back)
find $OUT -name '*.csv' -size +20M -delete
find $OUT -type f | head -50
