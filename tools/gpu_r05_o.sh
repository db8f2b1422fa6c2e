#!/bin/bash
# one-wave cores: tests of every path that runs them, the resident kernel's phase times, A/B of the config-1 stream and of the ragged batch replay
mkdir -p gpurun_out/r05o
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scan_resident or sequence or config1 or ragged or frozen or update_by or scan_grid or eskf" 2>&1 | tail -3 | tee gpurun_out/r05o/tests.txt
if [ -f leg-kilo_amd/libdbg_res.so ]; then
LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/libdbg_res.so timeout 600 python tools/stream_workload.py --kind vlp --scans 3 2>&1 | grep "core" | tail -2 | cut -c1-300 | tee gpurun_out/r05o/phases.txt
fi
L=gpurun_out/r05o/ab.txt
rm -f $L
for i in 1 2 3; do
for lib in leg-kilo_amd/liblegkilo_hip.so "$@"; do
  [ -f "$lib" ] || continue
  echo "== $lib" >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/stream_workload.py --kind vlp --scans 12 2>/dev/null | tail -1 | cut -c1-90 >> $L
  LEGKILO_HIP_LIB=$PWD/$lib timeout 600 python tools/ab_ragged.py 2>/dev/null | tail -2 >> $L
done; done
cat $L
