#!/bin/bash
# per-bucket launches: update launch of two workgroups (update | pool bookkeeping); all tests that stream, then A/B of the 5 x 20k stream
mkdir -p gpurun_out/r05r
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not overlay and not batch" 2>&1 | tail -3
L=gpurun_out/r05r/ab.txt
rm -f $L
for i in 1 2 3; do
for lib in leg-kilo_amd/liblegkilo_hip.so "$@"; do
  echo "== $lib" >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind 5 --scans 12 2>/dev/null | tail -1 | cut -c1-100 >> $L
done; done
cat $L
