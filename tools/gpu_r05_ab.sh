#!/bin/bash
# A/B of the stream workloads: shipped library against the ones given, interleaved, three rounds
mkdir -p gpurun_out/r05ab
L=gpurun_out/r05ab/ab.txt
rm -f $L
for i in 1 2 3; do
for lib in leg-kilo_amd/liblegkilo_hip.so "$@"; do
  [ -f "$lib" ] || continue
  echo "== $lib" >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind vlp --scans 12 2>/dev/null | tail -1 | cut -c1-90 >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 300 python tools/stream_small.py 2>/dev/null | tail -1 >> $L
done; done
cat $L
