#!/bin/bash
# the whole GPU suite in reverse order and in one shuffled order (poisoned allocations), one process each
mkdir -p gpurun_out/order
for ord in reverse shuffle:5; do
  LK_TEST_ORDER=$ord LEGKILO_POISON_POOLS=1 timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -s > gpurun_out/order/suite_$ord.txt 2>&1
  echo "== $ord"; grep -v "^  File" gpurun_out/order/suite_$ord.txt | grep -i "passed\|failed\|fault\|Aborted\|FAILED" | tail -n 8 | cut -c1-250
done
