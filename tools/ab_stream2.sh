#!/bin/bash
# same-box A/B of library builds on the stream-path extras of the bench line: tools/ab_stream2.sh <rounds> lib_a.so lib_b.so ...
R=$1; shift
for r in $(seq 1 $R); do
for v in "$@"; do
  LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/$v python bench.py --steps 10 --warmup 5 --cpu-sample 0 --stream-scans 24 --config1-scans 2048 --no-pcie --overlay-scans ${OVS:-0} --sustained-s 0 --cache-dir /tmp/lkcache 2>/dev/null > /tmp/ab_lib.json
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_lib.json").read().strip().splitlines()[-1])
e = d["extra"]
print(sys.argv[1], "stream", e.get("stream_ms_per_scan"), "stream51", e.get("stream51_ms_per_scan"), "config1 live", e.get("config1_live_stream_ms_per_scan"), "config1 batch", e.get("config1_scans_dev_ms_per_batch"), "overlay", e.get("overlay_ms_per_batch"), "value", d["value"])
PY
done
done
