#!/bin/bash
# A/B of the pipelined stream path (LEGKILO_SPEC=1, default) against the sequential order (LEGKILO_SPEC=0) on the three stream
# figures of the bench line: tools/ab_spec.sh [lib.so ...]   (libraries relative to leg-kilo_amd/, default the shipped one)
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(liblegkilo_hip.so)
for v in "${libs[@]}"; do
  for spec in 0 1 0 1; do
    LEGKILO_SPEC=$spec LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/$v python bench.py --steps 2 --warmup 1 --cpu-sample 0 --scans-per-gpu 32 --stream-scans 24 --config1-scans 64 --sustained-s 0 --no-pcie 2>/tmp/ab_spec.err > /tmp/ab_spec.json || tail -5 /tmp/ab_spec.err
    python - "$v" "$spec" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_spec.json").read().strip().splitlines()[-1])
e = d["extra"]
print(sys.argv[1], "SPEC", sys.argv[2], "stream ms/scan", e.get("stream_ms_per_scan"), "stream51", e.get("stream51_ms_per_scan"), "config1 live", e.get("config1_live_stream_ms_per_scan"), "stats", e.get("stream_pipeline_stats"))
PY
  done
done
