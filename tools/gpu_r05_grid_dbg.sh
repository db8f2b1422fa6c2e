mkdir -p gpurun_out/r05res
LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/libdbg_res.so timeout 600 python tools/stream_workload.py --kind 51 --scans 4 2>&1 | grep "^\[grid\]\|kind" | tail -4 | cut -c1-600 > gpurun_out/r05res/grid.txt
cat gpurun_out/r05res/grid.txt
