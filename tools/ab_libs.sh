#!/bin/bash
# A/B of library variants: tools/ab_libs.sh lib_a.so lib_b.so ...  (paths relative to leg-kilo_amd/)
for v in "$@"; do
  LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/$v python bench.py --steps 10 --warmup 2 --cpu-sample 0 --stream-scans 0 2>/dev/null > /tmp/ab_lib.json
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_lib.json").read().strip().splitlines()[-1])
print(sys.argv[1], "scans/s", d["value"], "ms/step", d["ms_per_step"], "residual_ms", d["roofline"]["launch_ms_single_stream_events"], "n_eff", d["extra"]["mean_n_effect"])
PY
done
