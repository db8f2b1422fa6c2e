#!/bin/bash
# Phase times inside the scan-resident stream kernel (debug build -DLK_DEBUG_RES): the bench line's config-1 live stream (tools/stream_workload.py
# --kind vlp) and a young map (tools/stream_small.py), then the libraries given as arguments beside the shipped one, interleaved.
mkdir -p gpurun_out/r05res
L=gpurun_out/r05res/log.txt
if [ -f leg-kilo_amd/libdbg_res.so ]; then
  echo "== debug build, vlp" >> $L
  LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/libdbg_res.so timeout 600 python tools/stream_workload.py --kind vlp --scans 8 2>&1 | grep -v "^map" | tail -20 >> $L
  echo "== debug build, young map" >> $L
  LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/libdbg_res.so timeout 300 python tools/stream_small.py 2>&1 | tail -7 >> $L
fi
for i in 1 2 3; do
for lib in leg-kilo_amd/liblegkilo_hip.so "$@"; do
  [ -f "$lib" ] || continue
  echo "== $lib" >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind vlp --scans 12 2>/dev/null | tail -1 | cut -c1-120 >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 300 python tools/stream_small.py 2>/dev/null | tail -1 >> $L
done; done
tail -80 $L
