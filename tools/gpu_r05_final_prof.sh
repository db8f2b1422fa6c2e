#!/bin/bash
# counters at HEAD: the graded kernel (cell order and shuffled), the overlay replay
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/gpu_prof_r05.sh r05d c865e573083c 2>&1 | tail -n 25
bash tools/gpu_prof_overlay_r05.sh r05d "stats fetch write sq" c865e573083c 2>&1 | tail -n 12
