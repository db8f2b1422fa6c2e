mkdir -p gpurun_out/r05res
timeout 600 python tools/stream_workload.py --kind vlp --scans 12 2>&1 | tail -1 > gpurun_out/r05res/redo.txt
cat gpurun_out/r05res/redo.txt
