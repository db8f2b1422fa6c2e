#!/bin/bash
mkdir -p gpurun_out/r05v
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlay" 2>&1 | tail -3
for i in 1 2 3; do
timeout 600 python tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32 --slots 1024 --reps 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_batch'], j['kernel_ms']['ov_insert_fallback'], j['kernel_ms']['ov_insert_apply'])"
done
