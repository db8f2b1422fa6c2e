#!/bin/bash
# one-XCD placement of the grid-resident stream kernel: bit identity and same-box timings against the any-XCD form, interleaved
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "scan_grid_kernel" 2>&1 | tail -2
for r in 1 2 3; do
for wg in 0 16; do
  for x in 1 0; do
    echo "== kind 51 XCD=$x WG=$wg"
    LEGKILO_GRIDSCAN_XCD=$x LEGKILO_GRIDSCAN_WG=$wg python tools/stream_workload.py --kind 51 --scans 24 --reps 3 2>/dev/null | tail -2
  done
done
done
echo "== kind 51 launches"
LEGKILO_GRIDSCAN=0 python tools/stream_workload.py --kind 51 --scans 24 --reps 3 2>/dev/null | tail -2
} > gpurun_out/r04_xcd.txt 2>&1
tail -40 gpurun_out/r04_xcd.txt
