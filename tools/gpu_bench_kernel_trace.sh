set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python $REPO/bench.py --cpu-sample 0 --no-pcie --sustained-s 0 --cache-dir /tmp/lkcache --steps 1 --warmup 0 --stream-scans 3 --overlay-scans 0 --config1-scans 0 > $OUT/warm.log 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04k -o bench -- python $REPO/bench.py --cpu-sample 0 --no-pcie --sustained-s 0 --cache-dir /tmp/lkcache --steps 20 --warmup 5 > $OUT/stats.log 2>&1 < /dev/null
echo rc=$?
f=$(find /tmp/r04k -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -n 14 $f | cut -c1-200
tail -c 300 $OUT/stats.log
