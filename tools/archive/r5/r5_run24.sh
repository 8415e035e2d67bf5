# residency of the polite 64x64 filter-gradient launches (LDS request per block), tuning build of conv_wgrad2.hip, alternating
mkdir -p gpurun_out/r5x
export P3D_LIB=$PWD/tools/ab/libp3dhip_wtune.so
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5x/$tag.json 2> gpurun_out/r5x/$tag.err; echo "$tag $(python3 -c "import json;print(json.loads(open('gpurun_out/r5x/$tag.json').read().strip().splitlines()[-1])['ms_per_step'])")"; }
for i in 1 2; do
run kb82_$i X=1
run kb55_$i P3D_WGRAD_LDS_KB=55
run kb48_$i P3D_WGRAD_LDS_KB=48
run kb110_$i P3D_WGRAD_LDS_KB=110
done
