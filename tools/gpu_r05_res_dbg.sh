mkdir -p gpurun_out/r05res
LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/libdbg_res.so timeout 600 python tools/stream_workload.py --kind vlp --scans 4 2>&1 | grep -v "^map" | tail -40 > gpurun_out/r05res/log.txt
cat gpurun_out/r05res/log.txt
