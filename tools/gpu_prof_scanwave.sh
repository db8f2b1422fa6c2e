#!/bin/bash
# PMC counters of the scan-resident replay kernel (lk_scan_wave_kernel, one wave per recorded scan) on the config-1 batch of
# tools/ab_ragged.py: two SQ passes, condensed on the box into gpurun_out/scanwave_pmc.txt.   usage: tools/gpu_prof_scanwave.sh [mode]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
MODE=${1:-plain}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d /tmp/sw_$name -o sw -- python $REPO/tools/ab_ragged.py $MODE > /tmp/sw_$name.log 2>&1 < /dev/null
  echo "$name rc=$?"
}
[ "${SW_PASSES:-12}" != 12 ] || pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
[ "${SW_PASSES:-12}" != 12 ] || pass 2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE
[ "${SW_PASSES:-12}" = 3 ] && pass 3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F64
[ "${SW_PASSES:-12}" = 4 ] && pass 4 SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_TMP_STALL_CSN SPI_RA_RES_STALL_CSN SPI_RA_BAR_CU_FULL_CSN
[ "${SW_PASSES:-12}" = 4 ] && pass 5 SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SPI_CSN_BUSY SPI_CSN_WAVE
python - > $OUT/scanwave_pmc.txt 2>&1 <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/sw_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lk_scan_wave' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k:28s} per-launch {v / max(n, 1):16.1f}  launches {n}")
PY
cat $OUT/scanwave_pmc.txt; tail -2 /tmp/sw_1.log
