#!/bin/bash
mkdir -p gpurun_out/r05s
LK_PROF_COMMIT=$1 timeout 1500 python tools/parity_all_slots.py --out gpurun_out/r05s/r05_parity_all_slots.json 2>&1 | tail -4 | cut -c1-600
