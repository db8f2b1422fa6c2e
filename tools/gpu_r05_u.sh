#!/bin/bash
# overlay replay: workgroups per slot of the (nearly always empty) fallback launch and of the apply launch
mkdir -p gpurun_out/r05u
L=gpurun_out/r05u/sweep.txt
rm -f $L
timeout 300 python tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32 --slots 64 --reps 1 > /dev/null 2>&1
for i in 1 2; do
for fb in 0 1 2 4; do
  echo "== FB_WG=$fb" >> $L
  LEGKILO_OV_FB_WG=$fb timeout 600 python tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32 --slots 1024 --reps 3 2>/dev/null | tail -1 | cut -c1-700 >> $L
done
for ap in 2 4; do
  echo "== APPLY_WG=$ap FB_WG=1" >> $L
  LEGKILO_OV_APPLY_WG=$ap LEGKILO_OV_FB_WG=1 timeout 600 python tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32 --slots 1024 --reps 3 2>/dev/null | tail -1 | cut -c1-700 >> $L
done
done
cat $L
