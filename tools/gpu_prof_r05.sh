#!/bin/bash
# Round 5 counters of the graded line's kernel (lk_residual_kernel<false,1,..>, frozen-map batch): the round-3 passes on the bench command
# (tools/gpu_prof_r03.sh -> profiles/latest_pmc.json), then the same kernel on the batch with a RANDOM order inside every bucket
# (bench.py --shuffle-main; fetch / write / tcc passes -> profiles/latest_shuffled_pmc.json), then a kernel trace of the plain bench command.
#   usage: tools/gpu_prof_r05.sh <tag> <commit>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05b}
COMMIT=${2:-unknown}
BASE="--cpu-sample 0 --config1-scans 0 --config2-scans 0 --config4-scans 0 --no-pcie --sustained-s 0 --overlay-scans 0 --shuffle-check 0 --cache-dir /tmp/lkcache"
cd $REPO
COMMON="$BASE" bash tools/gpu_prof_r03.sh $TAG "stats fetch write sq1 sq2 sq3 tcc" $COMMIT
cp gpurun_out/profiles_$TAG/latest_pmc.json gpurun_out/latest_pmc_$TAG.json 2>/dev/null
COMMON="$BASE --shuffle-main" bash tools/gpu_prof_r03.sh ${TAG}s "stats fetch write sq1 sq2 sq3 tcc" $COMMIT
python - <<PY
import json
j = json.load(open("gpurun_out/profiles_${TAG}s/latest_pmc.json"))
out = {"tag": j["tag"], "commit": j.get("commit"), "kernel_sources_sha16": j.get("kernel_sources_sha16"),
       "what": "bench.py --shuffle-main: the graded loop on the batch with a RANDOM order inside every time bucket (curvature kept); the passes of tools/gpu_prof_r03.sh (tools/gpu_prof_r05.sh)",
       "kernel": "lk_residual_kernel<false,1,true,false>", "points_per_launch": j["points_per_launch"], "hbm_bytes_per_launch": j["hbm_bytes_per_launch"],
       "hbm_bytes_per_point": j["hbm_bytes_per_point"], "tcc_req_per_point": j["tcc_req_per_point"], "tcc_hit_rate": j["tcc_hit_rate"],
       "valu_insts_per_wave": j.get("valu_insts_per_wave"), "wait_any_frac_of_wave_cycles": j.get("wait_any_frac_of_wave_cycles"), "avg_dur_us_profiled": j.get("profiled_dur_us")}
json.dump(out, open("gpurun_out/latest_shuffled_pmc_$TAG.json", "w"), indent=1)
print(out)
PY
ls gpurun_out/profiles_$TAG gpurun_out/profiles_${TAG}s 2>/dev/null
