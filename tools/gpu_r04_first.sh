#!/bin/bash
# round 4, first GPU pass: the overlay replay's parity tests, the readlane-gather probe + the library built with the readlane form,
# the full-size config-5 test, one bench line.  Everything is logged under gpurun_out/$TAG; no step stops the next.
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== overlay tests" | tee $OUT/steps.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -m gpu -k "test_batch_replay_overlay" > $OUT/overlay_tests.log 2>&1; echo "rc $?" | tee -a $OUT/steps.log
echo "== readlane probe" | tee -a $OUT/steps.log
timeout 120 tools/probes/readlane_gather_probe > $OUT/readlane_gather_probe.txt 2>&1; echo "rc $?" | tee -a $OUT/steps.log
echo "== insert tests on the library built with the readlane gather" | tee -a $OUT/steps.log
LEGKILO_HIP_LIB=$PWD/tools/probes/liblegkilo_hip_rl.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_update_points_bucket_and_insert or test_config3_full_size or test_map_update_surface or test_sequence_imu_mode or test_config3_soak_full_size or test_scan_resident_kernel_equals" > $OUT/readlane_lib_tests.log 2>&1; echo "rc $?" | tee -a $OUT/steps.log
echo "== config 5 full size" | tee -a $OUT/steps.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "test_config5_full_size_batch" > $OUT/config5_full.log 2>&1; echo "rc $?" | tee -a $OUT/steps.log
echo "== bench" | tee -a $OUT/steps.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc $?" | tee -a $OUT/steps.log
tail -c 600 $OUT/overlay_tests.log
tail -n 5 $OUT/readlane_gather_probe.txt
tail -n 3 $OUT/readlane_lib_tests.log
tail -n 3 $OUT/config5_full.log
tail -c 1500 $OUT/bench.err
