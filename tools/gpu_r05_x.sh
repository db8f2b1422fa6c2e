#!/bin/bash
mkdir -p gpurun_out/r05x
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05x/bench.json 2> gpurun_out/r05x/bench.err; echo "bench rc $?"; tail -n 3 gpurun_out/r05x/bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r05x/bench.json'))
e=j['extra']
print(j['value'], j['ms_per_step'], j['parity_check'].get('ok'), j['parity_check'].get('shuffled_resorted'))
print({k:v for k,v in e.items() if 'resort' in k or k.startswith('shuffled_in_bucket_ms')})
print(j.get('warnings'))
PY
