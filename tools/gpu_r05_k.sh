#!/bin/bash
# grid-resident kernel: workgroup-count sweep
mkdir -p gpurun_out/r05k
L=gpurun_out/r05k/sweep.txt
rm -f $L
for i in 1 2 3; do
for wg in 0 16 24 32 40 48; do
  echo "== WG=$wg" >> $L
  LEGKILO_GRIDSCAN_WG=$wg timeout 600 python tools/stream_workload.py --kind 51 --scans 10 2>/dev/null | tail -1 | cut -c1-100 >> $L
done
done
cat $L
