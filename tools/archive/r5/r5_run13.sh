mkdir -p gpurun_out/r5m
ulimit -c 0
timeout -k 10 300 python -X faulthandler -m pytest tests/test_gpu_schedule.py -m gpu -q -x > gpurun_out/r5m/sched_alone.log 2>&1; echo "schedule alone rc=$?"; tail -3 gpurun_out/r5m/sched_alone.log | cut -c1-300
AMD_LOG_LEVEL=1 timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_pinned.py tests/test_gpu_schedule.py -m gpu -q -x > gpurun_out/r5m/pinned_sched.log 2>&1; echo "pinned+schedule rc=$?"; grep -v "^  File" gpurun_out/r5m/pinned_sched.log | tail -12 | cut -c1-300
