#!/bin/bash
mkdir -p gpurun_out/poison
LEGKILO_POISON_POOLS=1 timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/poison/bench.json 2> gpurun_out/poison/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/poison/bench.json'))
print(j['value'], j['parity_check'].get('ok'), {k:(v.get('counts_equal'),v.get('n')) for k,v in j['parity_check'].items() if isinstance(v,dict)}, j['parity_check'].get('counts_equal'), j.get('warnings'))
PY
