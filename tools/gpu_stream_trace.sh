#!/bin/bash
# On the GPU box: kernel trace of the stream workload, pipelined and sequential, timeline of the last scan printed.
# tools/gpu_stream_trace.sh <kind 5|51|vlp> <tag>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
KIND=${1:-5}
TAG=${2:-r03}
NL=${3:-80}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in 1 0; do
  D=$OUT/trace_${TAG}_${KIND}_spec${spec}
  rm -rf $D
  timeout 300 rocprofv3 --kernel-trace -d $D -o t -- python $REPO/tools/stream_workload.py --kind $KIND --scans 4 --warm 4 --spec $spec > $OUT/trace_${TAG}_${KIND}_spec${spec}.log 2>&1
  tail -2 $OUT/trace_${TAG}_${KIND}_spec${spec}.log
  python $REPO/tools/trace_timeline.py $D $NL > $OUT/timeline_${TAG}_${KIND}_spec${spec}.txt 2>&1
  find $D -type f -size +30M -delete
done
