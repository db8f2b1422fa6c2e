#!/bin/bash
# TLB / L2 counters per kernel of a stream workload (names checked against rocprofv3 -L first):  tools/tlb_pmc.sh <kind 5|51> [tag] [all]
# Only the TCC / GRBM pass runs by default: the TCP_* passes (UTCL1 hits / misses, L2 request latencies) did not finish within 200 s
# each on the 90-launch workload in round 3 - pass "all" as third argument, and a long timeout, to try them.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
KIND=${1:-5}; TAG=${2:-tlb}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep Counter_Name | awk '{print $3}' | sort -u > /tmp/all_counters.txt
pick() { for c in "$@"; do grep -qx "$c" /tmp/all_counters.txt && echo -n "$c "; done; }
P1=$(pick TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum)
P2=$(pick TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum)
P3=$(pick TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE)
echo "pass1: $P1"; echo "pass2: $P2"; echo "pass3: $P3"
i=0
for pmc in "$P1" "$P2" "$P3"; do
  i=$((i+1)); D=/tmp/tlbp_$i; rm -rf $D
  [ -z "$pmc" ] && continue
  [ $i -lt 3 ] && [ "${3:-}" != "all" ] && continue
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $D -o t -- python $REPO/tools/stream_workload.py --kind $KIND --scans 8 --warm 6 --spec 0 > /tmp/tlbp_$i.log 2>&1
  tail -1 /tmp/tlbp_$i.log
done
python - <<'PY' | tee $OUT/${TAG}_${KIND}.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for i in (1, 2, 3):
    for f in glob.glob(f"/tmp/tlbp_{i}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:36]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if i == 1 and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); calls[k] += 1
for k, c in sorted(acc.items()):
    if not k.startswith("lk_"): continue
    g = lambda n: c.get(n, 0.0)
    print(f"{k:38s} calls {calls[k]:4d} utcl1 req {g('TCP_UTCL1_REQUEST_sum'):11.0f} miss {g('TCP_UTCL1_TRANSLATION_MISS_sum'):9.0f} ({g('TCP_UTCL1_TRANSLATION_MISS_sum') / max(g('TCP_UTCL1_REQUEST_sum'), 1):.4f})  "
          f"L2 read latency {g('TCP_TCC_READ_REQ_LATENCY_sum') / max(g('TCP_TCC_READ_REQ_sum'), 1):7.0f} cyc ({g('TCP_TCC_READ_REQ_sum'):10.0f} req)  write {g('TCP_TCC_WRITE_REQ_LATENCY_sum') / max(g('TCP_TCC_WRITE_REQ_sum'), 1):7.0f} cyc  "
          f"TCC hit {g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1):.3f} ea_rd {g('TCC_EA0_RDREQ_sum'):9.0f}  utcl2 busy {g('GRBM_UTCL2_BUSY') / max(g('GRBM_GUI_ACTIVE'), 1):.3f}")
PY
