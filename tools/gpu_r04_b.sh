#!/bin/bash
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in tiny scattered sectors; do
  echo "== overlay test $c" | tee -a $OUT/steps.log
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -m gpu -k "test_batch_replay_overlay and $c" > $OUT/overlay_$c.log 2>&1; echo "rc $?" | tee -a $OUT/steps.log
  tail -n 12 $OUT/overlay_$c.log | cut -c1-400
done
echo "== bench (overlay extra only)" | tee -a $OUT/steps.log
timeout 900 python bench.py --cpu-sample 24 --config1-scans 0 --stream-scans 0 --no-pcie --sustained-s 0 > $OUT/bench.json 2> $OUT/bench.err; echo "rc $?" | tee -a $OUT/steps.log
tail -c 800 $OUT/bench.err
python - <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r04b/bench.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("warnings"))
    print({k:v for k,v in d["extra"].items() if k.startswith("overlay")})
    print(d["parity_check"])
except Exception as e:
    print("no bench line:", e)
PY
