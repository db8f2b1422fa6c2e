#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "sort_by_voxel" 2>&1 | tail -15
