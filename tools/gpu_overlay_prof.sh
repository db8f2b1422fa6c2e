#!/bin/bash
# Overlay replay on the GPU box: slot sweep, kernel trace, PMC passes (one rocprofv3 run each, counters only with --kernel-trace).
#   tools/gpu_overlay_prof.sh <tag> [passes]   passes: subset of "sweep stats sq tcc lat fetch write r3"
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04d}
PASSES=${2:-"sweep stats sq tcc lat fetch write"}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W="python $REPO/tools/overlay_workload.py --cache-dir /tmp/lkcache --unique 32"
timeout 300 $W --slots 64 --reps 1 > $OUT/warm.log 2>&1 < /dev/null   # fills the input cache outside any profiler
pmc() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/ovp_$name
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/ovp_$name -o t -- $W --slots 1024 --reps 1 --no-profile > $OUT/pmc_$name.log 2>&1 < /dev/null
  echo "pmc $name rc=$?"
}
for p in $PASSES; do
  case $p in
    sweep) for s in 128 256 512 1024; do timeout 300 $W --slots $s --reps 3 2>/dev/null | tail -n 1 | tee -a $OUT/sweep.jsonl; done ;;
    stats) rm -rf /tmp/ovp_stats; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ovp_stats -o t -- $W --slots 1024 --reps 3 --no-profile > $OUT/stats.log 2>&1 < /dev/null
           f=$(find /tmp/ovp_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -n 16 $f ;;
    sq2)   pmc sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 ;;
    test)  timeout 600 python -m pytest $REPO/tests/test_gpu_parity.py -q -m gpu -k "test_batch_replay_overlay" -p no:cacheprovider > $OUT/overlay_tests.log 2>&1; echo "overlay tests rc $?"; tail -n 2 $OUT/overlay_tests.log ;;
    one)   timeout 300 $W --slots 1024 --reps 3 2>/dev/null | tail -n 1 | tee -a $OUT/sweep.jsonl ;;
    sq)    pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE ;;
    tcc)   pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE ;;
    lat)   pmc lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum ;;
    fetch) pmc fetch FETCH_SIZE ;;
    write) pmc write WRITE_SIZE ;;
    r3)    for v in ${R3V:-r3_readlane_nop r3_readlane_only21 r3_readlane_only9}; do
             LEGKILO_HIP_LIB=$REPO/tools/probes/liblegkilo_$v.so timeout 300 python -m pytest $REPO/tests/test_gpu_parity.py -q -m gpu -k "test_map_update_surface or test_update_points_bucket_and_insert" -p no:cacheprovider > $OUT/$v.log 2>&1; echo "$v rc $?"; tail -n 3 $OUT/$v.log | cut -c1-200
           done ;;
  esac
done
python - $OUT <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob("/tmp/ovp_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
res = {}
for k, c in sorted(acc.items()):
    if not k.startswith("lk_"): continue
    res[k] = dict(c); res[k]["dispatches_max"] = len(calls[k])
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for k, c in res.items():
    g = lambda n: c.get(n, 0.0)
    print(f"{k:46s} waves {g('SQ_WAVES'):9.0f} occupancy {g('SQ_WAVE_CYCLES') / max(g('SQ_BUSY_CYCLES'), 1):5.2f}/SIMD-ish  wait_inst {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.2f}  "
          f"valu {g('SQ_ACTIVE_INST_VALU') / max(g('SQ_BUSY_CYCLES'), 1):.3f}  TCC hit {g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1):.3f} req {g('TCC_REQ_sum'):.3g}  "
          f"L2 lat {g('TCP_TCC_READ_REQ_LATENCY_sum') / max(g('TCP_TCC_READ_REQ_sum'), 1):6.0f} cyc  utcl1 miss {g('TCP_UTCL1_TRANSLATION_MISS_sum') / max(g('TCP_UTCL1_REQUEST_sum'), 1):.4f}  "
          f"utcl2 busy {g('GRBM_UTCL2_BUSY') / max(g('GRBM_GUI_ACTIVE'), 1):.3f}  fetch {g('FETCH_SIZE') * 2 / 1e6:.1f} GB write {g('WRITE_SIZE') / 1e6:.1f} GB"
          + (f"  per wave: VALU {g('SQ_INSTS_VALU') / max(g('SQ_WAVES'), 1):.0f} (fp64 fma {g('SQ_INSTS_VALU_FMA_F64') / max(g('SQ_WAVES'), 1):.0f} mul {g('SQ_INSTS_VALU_MUL_F64') / max(g('SQ_WAVES'), 1):.0f} add {g('SQ_INSTS_VALU_ADD_F64') / max(g('SQ_WAVES'), 1):.0f}) SALU {g('SQ_INSTS_SALU') / max(g('SQ_WAVES'), 1):.0f} VMEM_RD {g('SQ_INSTS_VMEM_RD') / max(g('SQ_WAVES'), 1):.0f} LDS {g('SQ_INSTS_LDS') / max(g('SQ_WAVES'), 1):.0f}" if g('SQ_INSTS_VALU') else ""))
PY
