mkdir -p gpurun_out/r5q
timeout -k 10 300 tools/micro/bin/conv_chain 3 > gpurun_out/r5q/conv_chain.log 2>&1; echo "conv_chain rc=$?"; grep -A12 "^L1" gpurun_out/r5q/conv_chain.log | grep "^L1\|planner\|128x64  splits  1 "
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_determinism.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab/ab_libs.sh gpurun_out/r5q/ab8 3 "nopw product" --steps 30 --warmup 5
bash tools/ab/ab_libs.sh gpurun_out/r5q/ab32 2 "nopw product" --steps 10 --warmup 3 --batch 32
