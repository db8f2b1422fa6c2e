#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/gpu_prof_overlay_r05.sh r05e "stats fetch write sq" fdcf6091dca1 2>&1 | tail -n 6
