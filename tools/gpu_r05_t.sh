#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "timeout_restores" 2>&1 | tail -15
