#!/bin/bash
# same-box A/B of the three stream workloads: shipped library against the ones given
L=gpurun_out/r05ab/ab3.txt
mkdir -p gpurun_out/r05ab; rm -f $L
for i in 1 2 3; do
for lib in leg-kilo_amd/liblegkilo_hip.so "$@"; do
  [ -f "$lib" ] || continue
  echo "== $lib" >> $L
  for kind in 5 51 vlp; do
    LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind $kind --scans 12 2>/dev/null | tail -1 | cut -c1-62 >> $L
  done
done; done
cat $L
