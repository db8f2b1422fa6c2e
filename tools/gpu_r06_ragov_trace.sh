#!/bin/bash
# kernel trace of the recorded-run batch WITH insert (lk_batch_replay_overlay_ragged_dev, bench extra config1_overlay_ragged_*): per-kernel launch
# counts and durations -> is the 45 ms launch overhead or kernel time?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06_ragov
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-sample 0 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --stream-scans 0 --config2-scans 0 --config4-scans 0 --steps 2 --warmup 1 --cache-dir /tmp/lkcache"
python $REPO/bench.py $ARGS > $OUT/warm.json 2>/dev/null
rm -rf /tmp/ragov
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ragov -o t -- python $REPO/bench.py $ARGS > $OUT/bench.json 2> $OUT/trace.log
cp /tmp/ragov/*/*kernel_stats.csv $OUT/ 2>/dev/null || find /tmp/ragov -name '*kernel_stats.csv' -exec cp {} $OUT/ \;
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k:v for k,v in d["extra"].items() if k.startswith("config1_overlay")})
PY
head -30 $OUT/*kernel_stats.csv | cut -c1-160
