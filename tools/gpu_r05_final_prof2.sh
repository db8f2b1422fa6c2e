#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/gpu_prof_overlay_r05.sh r05f "stats fetch write sq" 2824230302a1 2>&1 | tail -n 6
