mkdir -p gpurun_out/r5j
timeout -k 10 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_determinism.py -m gpu -x -q > gpurun_out/r5j/net.log 2>&1; echo "net tests rc=$?"; tail -4 gpurun_out/r5j/net.log
tools/ab/ab_libs.sh gpurun_out/r5j/b8 3 "base product" --steps 30 --warmup 8
