#!/bin/bash
TAG=${1:-r04c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== overlay tests" | tee -a $OUT/steps.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "test_batch_replay_overlay" > $OUT/overlay_tests.log 2>&1; echo "rc $?" | tee -a $OUT/steps.log
grep -E "passed|failed|Error|overlay high" $OUT/overlay_tests.log | cut -c1-300
B="python bench.py --cpu-sample 24 --config1-scans 0 --stream-scans 0 --no-pcie --sustained-s 0 --cache-dir /tmp/lkcache"
for w in 2 3 4; do
  echo "== bench overlay, root kernel at $w waves/SIMD" | tee -a $OUT/steps.log
  LEGKILO_OV_ROOT_WAVES=$w timeout 900 $B > $OUT/bench_w$w.json 2> $OUT/bench_w$w.err; echo "rc $?" | tee -a $OUT/steps.log
  python - $OUT/bench_w$w.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e=d["extra"]; print(d["value"], e.get("overlay_ms_per_batch"), e.get("overlay_scans_per_s"), e.get("overlay_kernel_ms_per_batch"), d["parity_check"].get("overlay"), e.get("overlay_error"))
except Exception as ex:
    print("no bench line:", ex)
PY
done
for wg in 8 16; do
  echo "== bench overlay, $wg workgroups per slot in the per-root passes" | tee -a $OUT/steps.log
  LEGKILO_OV_WG_PER_SLOT=$wg timeout 900 $B --cpu-sample 0 > $OUT/bench_wg$wg.json 2> $OUT/bench_wg$wg.err; echo "rc $?" | tee -a $OUT/steps.log
  python - $OUT/bench_wg$wg.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e=d["extra"]; print(d["value"], e.get("overlay_ms_per_batch"), e.get("overlay_kernel_ms_per_batch"), e.get("overlay_error"))
except Exception as ex:
    print("no bench line:", ex)
PY
done
echo "== round-3 tree (bdcb3a3) built with the readlane gather / shfl gather / readlane without LDS promotion: insert tests" | tee -a $OUT/steps.log
for v in r3_readlane r3_shfl r3_readlane_nolds; do
  LEGKILO_HIP_LIB=$PWD/tools/probes/liblegkilo_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_update_points_bucket_and_insert or test_config3_full_size or test_map_update_surface or test_sequence_imu_mode or test_config3_soak_full_size" > $OUT/$v.log 2>&1; echo "$v rc $?" | tee -a $OUT/steps.log
  tail -n 4 $OUT/$v.log | cut -c1-300
done
