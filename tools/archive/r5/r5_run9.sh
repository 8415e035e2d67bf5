mkdir -p gpurun_out/r5i
timeout -k 10 900 python -m pytest tests/test_gpu_full.py -m gpu -x -q -k "bottleneck_gradients_at_batch8" > gpurun_out/r5i/blocks.log 2>&1; echo "block tests rc=$?"; tail -3 gpurun_out/r5i/blocks.log
tools/ab/ab_libs.sh gpurun_out/r5i/b8 3 "base product" --steps 30 --warmup 8
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernels > gpurun_out/r5i/bench.json 2> gpurun_out/r5i/kernels.txt
grep "coop" gpurun_out/r5i/kernels.txt
