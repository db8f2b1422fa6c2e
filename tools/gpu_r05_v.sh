#!/bin/bash
mkdir -p gpurun_out/r05v
for i in 1 2; do
for fb in 32 0 128 256 512; do
echo "== FB_WG=$fb"
LEGKILO_OV_FB_WG=$fb timeout 600 python tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32 --slots 1024 --reps 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_batch'], j['kernel_ms']['ov_insert_fallback'], j['kernel_ms']['ov_insert_apply'])"
done; done
