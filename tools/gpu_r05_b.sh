#!/bin/bash
# Round 5: overlay tests + overlay workload (per-kernel times); optional env A/B list in $2 ("VAR=val VAR2=val;VAR=val...")
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r05b}
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "overlay" -p no:cacheprovider > $OUT/pytest_overlay.log 2>&1; echo "pytest overlay rc $?"; tail -n 12 $OUT/pytest_overlay.log | cut -c1-400
W="python $REPO/tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32"
timeout 300 $W --slots 64 --reps 1 > $OUT/warm.log 2>&1 < /dev/null
IFS=';' read -ra VARS <<< "${2:-LK_NONE=1}"
for v in "${VARS[@]}"; do
  echo "== $v"
  env $v timeout 300 $W --slots 1024 --reps 3 2>$OUT/ov.err | tail -n 1 | tee -a $OUT/ov.jsonl
  tail -n 3 $OUT/ov.err | cut -c1-300
done
