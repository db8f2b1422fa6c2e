#!/bin/bash
# Instruction-cache counters per kernel of a stream workload:  tools/icache_pmc.sh <kind 5|51|vlp> [tag]
# (two --pmc passes with --kernel-trace only; sums over all dispatches of a kernel)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
KIND=${1:-5}; TAG=${2:-icache}
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1)); D=/tmp/icp_$i; rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $D -o t -- python $REPO/tools/stream_workload.py --kind $KIND --scans 8 --warm 6 --spec 0 > /tmp/icp_$i.log 2>&1
  tail -1 /tmp/icp_$i.log
done
python - <<'PY' | tee $OUT/${TAG}_${KIND}.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for d in ("/tmp/icp_1", "/tmp/icp_2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if d.endswith("1") and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); calls[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if not k.startswith("lk_"): continue
    req, hit, mis, dup = (c.get(n, 0) for n in ("SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"))
    print(f"{k:42s} calls {calls[k]:5d}  icache req {req:12.0f}  hit {hit / max(req, 1):.3f}  miss {mis:10.0f} (+dup {dup:10.0f})  "
          f"ifetch latency {c.get('SQ_IFETCH_LEVEL', 0) / max(c.get('SQ_IFETCH', 1), 1):7.1f} cyc  wait_inst_any/wave_cycles {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
