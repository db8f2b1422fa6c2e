#!/bin/bash
mkdir -p gpurun_out/r05c4
timeout 1500 python tools/config4_full.py --scans 600 --out gpurun_out/r05c4/r05_config4_full.json 2>&1 | tail -3 | cut -c1-1500
