#!/bin/bash
# grid-resident kernel without the fallback body: its bit-identity tests, then A/B of the 51-bucket and 5 x 20k streams
mkdir -p gpurun_out/r05j
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "scan_grid or scan_resident or config3" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/r05j/tests.txt
cat gpurun_out/r05j/tests.txt
L=gpurun_out/r05j/ab.txt
rm -f $L
for i in 1 2 3; do
for lib in leg-kilo_amd/liblegkilo_hip.so "$@"; do
  [ -f "$lib" ] || continue
  echo "== $lib" >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind 51 --scans 10 2>/dev/null | tail -1 | cut -c1-300 >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind 5 --scans 10 2>/dev/null | tail -1 | cut -c1-100 >> $L
done; done
cat $L
