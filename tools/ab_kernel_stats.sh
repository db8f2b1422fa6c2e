#!/bin/bash
# Same-box A/B of environment switches on the stream workload by KERNEL durations (rocprofv3 --kernel-trace --stats), which are
# far less noisy than end-to-end times:  tools/ab_kernel_stats.sh <kind 5|51|vlp> "ENV=.. ENV2=.." "ENV=.." ...
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
KIND=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  D=/tmp/abks_$i; rm -rf $D
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python $REPO/tools/stream_workload.py --kind $KIND --scans 12 --warm 6 --spec 0 > /tmp/abks_$i.log 2>&1
  echo "== $v   $(tail -1 /tmp/abks_$i.log)"
  python - $D <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if not n.startswith(("lk_", "void lk_")): continue
    tot += float(r["TotalDurationNs"])
    print(f"   {n[:60]:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.2f} us  max {float(r['MaxNs'])/1e3:8.2f}")
print(f"   total lk kernels {tot/1e6:.2f} ms")
PY
done
