#!/bin/bash
# Round 5: kernel-trace timelines of the three single-stream shapes (sequential path), last scan printed
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for KIND in 5 51 vlp; do
  D=/tmp/trace_$KIND
  rm -rf $D
  timeout 300 rocprofv3 --kernel-trace -d $D -o t -- python $REPO/tools/stream_workload.py --kind $KIND --scans 4 --warm 20 > $OUT/trace_$KIND.log 2>&1
  tail -1 $OUT/trace_$KIND.log | cut -c1-200
  python $REPO/tools/trace_timeline.py $D ${1:-70} > $OUT/timeline_$KIND.txt 2>&1
done
head -45 $OUT/timeline_5.txt
