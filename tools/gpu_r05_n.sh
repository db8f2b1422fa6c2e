#!/bin/bash
# 5 x 20 000-point scans through the grid-resident kernel (LEGKILO_GRIDSCAN=2) against the per-bucket launches
mkdir -p gpurun_out/r05n2
L=gpurun_out/r05n2/grid5.txt
rm -f $L
for i in 1 2; do
  echo "== launches" >> $L
  timeout 600 python tools/stream_workload.py --kind 5 --scans 10 2>/dev/null | tail -1 | cut -c1-100 >> $L
  for wg in 48 64 96 128; do
    echo "== grid WG=$wg" >> $L
    LEGKILO_GRIDSCAN=2 LEGKILO_GRIDSCAN_WG=$wg timeout 600 python tools/stream_workload.py --kind 5 --scans 10 2>/dev/null | tail -1 | cut -c1-100 >> $L
  done
done
cat $L
