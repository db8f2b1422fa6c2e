#!/bin/bash
# resident-kernel tests, debug phase profile, then A/B of the stream workloads (shipped library against the ones given)
mkdir -p gpurun_out/r05i
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scan_resident or sequence or config1 or ragged" 2>&1 | tail -5 > gpurun_out/r05i/tests.txt
cat gpurun_out/r05i/tests.txt
if [ -f leg-kilo_amd/libdbg_res.so ]; then
LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/libdbg_res.so timeout 600 python tools/stream_workload.py --kind vlp --scans 3 2>&1 | grep "filter wave\|core\|insert wave" | tail -4 | cut -c1-300 | tee gpurun_out/r05i/phases.txt
fi
L=gpurun_out/r05i/ab.txt
for i in 1 2 3; do
for lib in leg-kilo_amd/liblegkilo_hip.so "$@"; do
  [ -f "$lib" ] || continue
  echo "== $lib" >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind vlp --scans 12 2>/dev/null | tail -1 | cut -c1-90 >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 300 python tools/stream_small.py 2>/dev/null | tail -1 >> $L
done; done
cat $L
