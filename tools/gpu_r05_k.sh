#!/bin/bash
# grid-resident kernel: workgroup-count sweep and the two-waves-per-SIMD build
mkdir -p gpurun_out/r05k
L=gpurun_out/r05k/sweep.txt
rm -f $L
for i in 1 2; do
for wg in 0 8 16 24 32; do
  echo "== WG=$wg" >> $L
  LEGKILO_GRIDSCAN_WG=$wg timeout 600 python tools/stream_workload.py --kind 51 --scans 10 2>/dev/null | tail -1 | cut -c1-100 >> $L
done
for lib in "$@"; do
  echo "== $lib" >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind 51 --scans 10 2>/dev/null | tail -1 | cut -c1-100 >> $L
done
done
cat $L
