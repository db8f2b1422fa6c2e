#!/bin/bash
mkdir -p gpurun_out/r05soak
timeout 1500 python tools/soak_long.py 60 2>&1 | tail -8 | tee gpurun_out/r05soak/soak.txt | cut -c1-300
