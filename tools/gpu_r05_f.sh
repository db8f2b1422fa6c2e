#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -n 4
for v in "LK_NONE=1" "LEGKILO_TAIL_MERGE=0" "LEGKILO_SPIN_WAIT=0" "LK_NONE=1" "LEGKILO_TAIL_MERGE=0"; do
  echo "== $v"
  env $v python bench.py --steps 20 --warmup 5 --cpu-sample 0 --overlay-scans 0 --shuffle-check 0 --no-pcie --sustained-s 0 --config1-scans 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j['extra']; print({k:e[k] for k in e if k.startswith(('stream_ms','stream51_ms','config1_live'))})"
done
