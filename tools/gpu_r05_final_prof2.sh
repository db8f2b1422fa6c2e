#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/gpu_prof_overlay_r05.sh r05g "stats fetch write sq" 96301b66992c 2>&1 | tail -n 6
