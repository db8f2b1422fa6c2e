#!/bin/bash
# same-box timing of overlay library variants: tools/gpu_ov_attr.sh lib_a.so lib_b.so ... (paths relative to leg-kilo_amd/); per-kernel ms of one replay
mkdir -p gpurun_out
{
for v in "$@"; do
  echo "== $v"
  LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/$v timeout 300 python tools/overlay_workload.py --slots ${SLOTS:-512} --unique 32 --reps 2 --cache-dir /tmp/lkcache 2>&1 | tail -1 | cut -c1-900
done
} > gpurun_out/r04_ov_attr.txt 2>&1
cat gpurun_out/r04_ov_attr.txt
