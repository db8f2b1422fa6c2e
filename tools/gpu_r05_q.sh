#!/bin/bash
# young map (first frame only): 51-bucket scans through the grid-resident kernel (relaunches at fallback items) against per-bucket launches
for i in 1 2; do
for m in 1 0; do
  echo "== LEGKILO_GRIDSCAN=$m"
  LEGKILO_GRIDSCAN=$m timeout 600 python tools/stream_workload.py --kind 51 --warm 1 --scans 8 2>/dev/null | tail -1 | cut -c1-330
done; done
