mkdir -p gpurun_out/r5l
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r5l/all_tests.log 2>&1; echo "all tests rc=$?"; tail -5 gpurun_out/r5l/all_tests.log
