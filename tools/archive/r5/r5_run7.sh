mkdir -p gpurun_out/r5g
timeout -k 10 900 python -m pytest tests/test_gpu_full.py -m gpu -x -q -k "bottleneck_gradients_at_batch8 or bottlenecks_at_batch8" > gpurun_out/r5g/blocks.log 2>&1; echo "block tests rc=$?"; tail -3 gpurun_out/r5g/blocks.log
tools/ab/ab_libs.sh gpurun_out/r5g/b8 3 "base product" --steps 30 --warmup 8
tools/ab/ab_libs.sh gpurun_out/r5g/b32 1 "base product" --batch 32 --steps 6 --warmup 2
P3D_WRITE_SCHEDULE_GOLDEN=gpurun_out/r5g/sched timeout -k 10 300 python -m pytest tests/test_gpu_schedule.py -m gpu -q > gpurun_out/r5g/sched.log 2>&1; echo "sched rc=$?"
P3D_MEASURE_GATES=gpurun_out/r5g/gates.json timeout -k 10 1100 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_schedule.py > gpurun_out/r5g/all_tests.log 2>&1; echo "all tests rc=$?"; tail -8 gpurun_out/r5g/all_tests.log
