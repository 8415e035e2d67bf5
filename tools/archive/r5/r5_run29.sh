mkdir -p gpurun_out/r5l gpurun_out/r5ac
P3D_WRITE_SCHEDULE_GOLDEN=gpurun_out/r5ac/sched timeout -k 10 300 python -m pytest tests/test_gpu_schedule.py -m gpu -x -q 2>&1 | tail -1
cp gpurun_out/r5ac/sched/*.txt tests/golden/
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r5l/all_tests.log 2>&1; echo "all tests rc=$?"; tail -3 gpurun_out/r5l/all_tests.log
bash tools/ab/ab_libs.sh gpurun_out/r5ac/ab8 2 "base product" --steps 30 --warmup 5
bash tools/ab/ab_libs.sh gpurun_out/r5ac/ab32 2 "base product" --steps 10 --warmup 3 --batch 32
bash tools/ab/ab_libs.sh gpurun_out/r5ac/ab224 1 "base product" --steps 8 --warmup 2 --frames 32 --size 224
