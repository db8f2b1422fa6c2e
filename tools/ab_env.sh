#!/bin/bash
# A/B of environment switches / library builds on the batch step:  tools/ab_env.sh "LEGKILO_GRID=0" "LEGKILO_GRID=1" "LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/lib_x.so" ...
# Each variant runs bench.py (1024 distinct scans, generated once and cached under /tmp for the session) with the extras off.
ARGS="${AB_EXTRA:-} --steps ${AB_STEPS:-20} --warmup 3 --cpu-sample ${AB_CPU_SAMPLE:-0} --stream-scans 0 --config1-scans 0 --no-pcie --sustained-s 0 --cache-dir /tmp/lkcache"
for v in "$@"; do
  for rep in 1 2; do
    env $v python bench.py $ARGS 2>/dev/null > /tmp/ab_env.json
    python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_env.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[1]:50s} scans/s {d['value']:10.1f}  ms/step {d['ms_per_step']:.4f}  launch_ms(events) {r['launch_ms_single_stream_events']:.4f}  ps/pt {r['ps_per_point']:.2f}  n_eff {d['extra']['mean_n_effect']:.2f}  parity {d['parity_check'] and (d['parity_check']['ok'], d['parity_check']['counts_equal'], d['parity_check']['max_pos_delta_m'])}")
PY
  done
done
