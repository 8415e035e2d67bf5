mkdir -p gpurun_out/r5u
P3D_WRITE_SCHEDULE_GOLDEN=gpurun_out/r5u/sched timeout -k 10 300 python -m pytest tests/test_gpu_schedule.py -m gpu -x -q 2>&1 | tail -2
bash tools/ab/ab_libs.sh gpurun_out/r5u/ab8 3 "base product" --steps 30 --warmup 5
bash tools/ab/ab_libs.sh gpurun_out/r5u/ab32 2 "base product" --steps 10 --warmup 3 --batch 32
bash tools/ab/ab_libs.sh gpurun_out/r5u/ab224 2 "base product" --steps 8 --warmup 2 --frames 32 --size 224
