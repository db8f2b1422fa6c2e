#!/bin/bash
# kernel-trace timelines of the config-1 live stream and the 51-bucket stream at HEAD (what is between two scans' kernels)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05t2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for KIND in ${KINDS:-vlp 51}; do
  D=/tmp/trace_$KIND
  rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $D -o t -- python $REPO/tools/stream_workload.py --kind $KIND --scans 4 --warm 6 > $OUT/trace_$KIND.log 2>&1
  tail -1 $OUT/trace_$KIND.log | cut -c1-200
  python $REPO/tools/trace_timeline.py $D 40 > $OUT/timeline_$KIND.txt 2>&1
  tail -25 $OUT/timeline_$KIND.txt
done
