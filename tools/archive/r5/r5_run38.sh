# final soak: the full GPU suite, 300 back-to-back steps, smoke, the two-rank rehearsal of the launch path on one GPU
mkdir -p gpurun_out/r5al
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r5al/all_tests.log 2>&1; echo "all tests rc=$?"; tail -2 gpurun_out/r5al/all_tests.log
timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2> gpurun_out/r5al/long.err | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
P3D_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r5al/reh.err | cut -c1-200
