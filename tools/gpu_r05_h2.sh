#!/bin/bash
mkdir -p gpurun_out/r05h
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "scan_resident" 2>&1 | tail -25 > gpurun_out/r05h/tests2.txt
cat gpurun_out/r05h/tests2.txt
