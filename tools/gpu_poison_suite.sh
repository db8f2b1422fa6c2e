#!/bin/bash
# The whole GPU suite with every device allocation of the library poisoned (LEGKILO_POISON_POOLS=1: 0x5a bytes instead of the allocator's zeros):
# a kernel that trusts memory nobody has written fails here in every run instead of in one unlucky process.
mkdir -p gpurun_out/poison
LEGKILO_POISON_POOLS=1 timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -s > gpurun_out/poison/suite.txt 2>&1
grep -v "^  File" gpurun_out/poison/suite.txt | grep -i "passed\|failed\|fault\|error\|Aborted\|FAILED" | tail -n 20 | cut -c1-300
