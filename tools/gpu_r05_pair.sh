#!/bin/bash
# Round 5 experiment: two tiles per wave in the batch residual kernel (LEGKILO_RES_PAIR=1), 3 and 4 waves per SIMD, against the shipped kernel; the bench's own parity check validates every run
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
B="python bench.py --steps 40 --warmup 10 --stream-scans 0 --config1-scans 0 --no-pcie --overlay-scans 0 --shuffle-check 0 --sustained-s 0.6 --cpu-sample 48 --cache-dir /tmp/lkcache"
run() { env "$@" $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scans/s', j['value'], 'ms/step', j['ms_per_step'], 'sustained', j['extra'].get('sustained_scans_per_s'), 'parity', j['parity_check']['ok'], j['parity_check']['counts_equal'], j['parity_check']['max_pos_delta_m'])"; }
for r in 1 2; do
echo "base"; run LK_NONE=1
echo "pair3"; run LEGKILO_RES_PAIR=1
echo "pair4"; run LEGKILO_RES_PAIR=1 LEGKILO_HIP_LIB=$REPO/leg-kilo_amd/liblegkilo_pw4.so
done
