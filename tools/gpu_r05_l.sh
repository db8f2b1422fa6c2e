#!/bin/bash
mkdir -p gpurun_out/r05l
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scan_grid" 2>&1 | tail -3 > gpurun_out/r05l/tests.txt
cat gpurun_out/r05l/tests.txt
LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/libdbg_res.so timeout 600 python tools/stream_workload.py --kind 51 --scans 3 2>&1 | grep "^\[grid\]" | tail -1 | cut -c1-600
L=gpurun_out/r05l/ab.txt
rm -f $L
for i in 1 2 3; do
for lib in leg-kilo_amd/liblegkilo_hip.so "$@"; do
  echo "== $lib" >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind 51 --scans 10 2>/dev/null | tail -1 | cut -c1-100 >> $L
done; done
cat $L
