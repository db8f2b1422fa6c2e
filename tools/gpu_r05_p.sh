#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "scan_grid" 2>&1 | grep -v "^$" | tail -8
