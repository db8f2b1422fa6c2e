#!/bin/bash
# Resident kernel without the fallback body, team of 7: parity tests of the resident kernel, then A/B of the stream workloads.
mkdir -p gpurun_out/r05h
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "scan_resident or config1 or sequence" 2>&1 | tail -25 > gpurun_out/r05h/tests.txt
cat gpurun_out/r05h/tests.txt
L=gpurun_out/r05h/ab.txt
for i in 1 2 3; do
for lib in leg-kilo_amd/liblegkilo_hip.so "$@"; do
  [ -f "$lib" ] || continue
  echo "== $lib" >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind vlp --scans 12 2>/dev/null | tail -1 | cut -c1-120 >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 300 python tools/stream_small.py 2>/dev/null | tail -1 >> $L
done; done
cat $L
