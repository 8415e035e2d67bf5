# grouped filter gradients of stages 2-3 on larger tiles (tuning build of conv_wgrad2.hip), same box, alternating
mkdir -p gpurun_out/r5s
export P3D_LIB=$PWD/tools/ab/libp3dhip_wtune.so
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5s/$tag.json 2> gpurun_out/r5s/$tag.err; echo "$tag $(python3 -c "import json;print(json.loads(open('gpurun_out/r5s/$tag.json').read().strip().splitlines()[-1])['ms_per_step'])")"; }
for i in 1 2; do
run base_$i X=1
run t64x128_$i P3D_TUNE_WGRAD_GROUP_TILE=1
run t128x128_$i P3D_TUNE_WGRAD_GROUP_TILE=2
run t128x64_$i P3D_TUNE_WGRAD_GROUP_TILE=3
done
timeout -k 10 600 env P3D_TUNE_WGRAD_GROUP_TILE=2 python -m pytest tests/test_gpu_net.py -m gpu -x -q 2>&1 | tail -2
