#!/bin/bash
# A/B of library variants on the single-stream path (config 3): tools/ab_stream.sh lib_a.so lib_b.so ...
for v in "$@"; do
  for i in 1 2 3 4; do
    LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/$v python bench.py --steps 2 --warmup 1 --cpu-sample 0 --scans-per-gpu 32 --stream-scans 24 2>/dev/null > /tmp/ab_st.json
    python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_st.json").read().strip().splitlines()[-1])
print(sys.argv[1], "stream ms/scan", d["extra"]["stream_ms_per_scan"], "scans/s", d["extra"]["stream_scans_per_s"])
PY
  done
done
