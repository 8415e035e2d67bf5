# persistent grid for the filter-gradient launches that run one block per CU (P3D_TUNE_WGRAD_PERSIST: 0 = every pair its own block)
mkdir -p gpurun_out/r5ai
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_determinism.py -m gpu -x -q 2>&1 | tail -2
export P3D_LIB=$PWD/tools/ab/libp3dhip_wtune.so
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5ai/$tag.json 2> gpurun_out/r5ai/$tag.err; echo "$tag $(python3 -c "import json;d=json.loads(open('gpurun_out/r5ai/$tag.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"; }
for i in 1 2 3; do
run off_$i P3D_TUNE_WGRAD_PERSIST=0
run p256_$i P3D_TUNE_WGRAD_PERSIST=256
run p128_$i P3D_TUNE_WGRAD_PERSIST=128
run p192_$i P3D_TUNE_WGRAD_PERSIST=192
done
