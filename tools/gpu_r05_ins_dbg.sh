mkdir -p gpurun_out/r05res
LEGKILO_HIP_LIB=$PWD/leg-kilo_amd/libdbg_ins.so timeout 600 python tools/stream_workload.py --kind vlp --scans 6 2>&1 | sed -n '/timed stream/,$p' | tail -60 > gpurun_out/r05res/ins.txt
cat gpurun_out/r05res/ins.txt
