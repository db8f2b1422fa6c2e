#!/bin/bash
# MFMA instruction counters of the config-1 batch (tools/ab_ragged.py: lk_scan_wave_kernel) for the shipped library and for the
# -DLK_MFMA_COV=1 build (covariance update of the one-wave update core on v_mfma_f64_16x16x4_f64): one rocprofv3 --pmc pass each.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in liblegkilo_hip.so liblegkilo_mfmacov.so; do
  rm -rf /tmp/mc_$v
  LEGKILO_HIP_LIB=$REPO/leg-kilo_amd/$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/mc_$v -o mc -- python $REPO/tools/ab_ragged.py plain > /tmp/mc_$v.log 2>&1 < /dev/null
  echo "== $v rc=$? $(tail -n 1 /tmp/mc_$v.log | cut -c1-200)"
  python - $v <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"/tmp/mc_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(acc.items()):
    if "scan_wave" in k or "update_wave" in k or "scan_stream" in k:
        w = max(c.get("SQ_WAVES", 0), 1)
        print(f"  {k:40s} per wave: MFMA f64 MOPS {c.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0) / w:10.1f}  VALU {c.get('SQ_INSTS_VALU', 0) / w:10.0f}  LDS {c.get('SQ_INSTS_LDS', 0) / w:10.0f}  waves {w:.0f}")
PY
done
