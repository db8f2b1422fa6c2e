#!/bin/bash
# same-box A/B of library builds on the graded step, alternating: tools/ab_libs2.sh <rounds> lib_a.so lib_b.so ...  (paths relative to leg-kilo_amd/)
R=$1; shift
for r in $(seq 1 $R); do
for v in "$@"; do
  LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/$v python bench.py --steps 40 --warmup 30 --cpu-sample 0 --stream-scans 0 --config1-scans 0 --no-pcie --overlay-scans 0 --sustained-s 0.6 --cache-dir /tmp/lkcache 2>/dev/null > /tmp/ab_lib.json
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_lib.json").read().strip().splitlines()[-1])
print(sys.argv[1], "scans/s", d["value"], "ms/step", d["ms_per_step"], "sustained", d["extra"].get("sustained_scans_per_s"), "n_eff", d["extra"]["mean_n_effect"])
PY
done
done
