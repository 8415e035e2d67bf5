mkdir -p gpurun_out/r5o
timeout -k 10 300 tools/micro/bin/conv_chain 3 > gpurun_out/r5o/conv_chain.log 2>&1; echo "conv_chain rc=$?"; grep -A12 "^L1" gpurun_out/r5o/conv_chain.log | grep "^L1\|planner\|128x64  splits  1 \|64x64  splits  1 "
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
