#!/bin/bash
# A/B of the XCD-aware batch residual launch (LEGKILO_XCDMAP): the bench headline and the config-2 rows launch, interleaved on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_xcd
mkdir -p $OUT
B="python $REPO/bench.py --cache-dir /tmp/lkcache --cpu-sample 24 --config1-scans 0 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0"
$B > /dev/null 2>&1   # fills the cache
for rep in 1 2; do
  for m in 0 1; do
    LEGKILO_XCDMAP=$m $B > $OUT/bench_$m.$rep.json 2> $OUT/bench_$m.$rep.err
    python - <<PY
import json
d=json.loads(open("$OUT/bench_$m.$rep.json").read().strip().splitlines()[-1])
print("XCDMAP=$m rep $rep: value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity_check"]["ok"], d["parity_check"]["counts_equal"])
PY
    LEGKILO_XCDMAP=$m python $REPO/tools/config2_workload.py --cache-dir /tmp/lkcache --slots 256 --reps 8 --no-calib 2>/dev/null | tail -1 | cut -c1-160
  done
done
