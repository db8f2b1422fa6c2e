#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -n 4
for kind in 5 51 vlp; do timeout 300 python tools/stream_workload.py --kind $kind --scans 12 --warm 20 --reps 3 2>/dev/null | tail -n 1 | cut -c1-200; done
