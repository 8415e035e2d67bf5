set -o pipefail
mkdir -p gpurun_out/r5a
cd $GRAFT_REPO_ROOT
for cfg in "5 32 0" "5 8 0" "5 1 0" "5 32 1"; do
  echo "== gated_chain $cfg" >> gpurun_out/r5a/gated_chain.log
  timeout -k 10 120 tools/micro/bin/gated_chain $cfg >> gpurun_out/r5a/gated_chain.log 2>&1 || { echo "gated_chain $cfg failed rc=$?" >> gpurun_out/r5a/gated_chain.log; break; }
done
tail -30 gpurun_out/r5a/gated_chain.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5a/bench_b8.json 2> gpurun_out/r5a/bench_b8.err && tail -c 600 gpurun_out/r5a/bench_b8.json | head -c 400; echo
python bench.py --batch 32 --steps 6 --warmup 2 --no-cpu-baseline --per-layer gpurun_out/r5a/r05_per_layer_b32.csv --dump-launches gpurun_out/r5a/launches_b32.csv > gpurun_out/r5a/bench_b32.json 2> gpurun_out/r5a/bench_b32.err; echo "b32 rc=$?"
P3D_MEASURE_GATES=gpurun_out/r5a/gates.json timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_full.py -m gpu -x -q -k "gn_head" > gpurun_out/r5a/gn_head_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r5a/gn_head_tests.log
