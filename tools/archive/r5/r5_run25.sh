# which filter-gradient launches should be polite (one block per CU) at 32 clips: P3D_WGRAD_POLITE_ROWS, tuning build of conv_wgrad2.hip
mkdir -p gpurun_out/r5y
export P3D_LIB=$PWD/tools/ab/libp3dhip_wtune.so
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 32 > gpurun_out/r5y/$tag.json 2> gpurun_out/r5y/$tag.err; echo "$tag $(python3 -c "import json;d=json.loads(open('gpurun_out/r5y/$tag.json').read().strip().splitlines()[-1]);r=d['roofline'];print(d['ms_per_step'],r['kernel'],r['avg_launch_us'],r['frac'])")"; }
for i in 1 2; do
run rows8192_$i X=1
run rows2048_$i P3D_WGRAD_POLITE_ROWS=2048
run rows0_$i P3D_WGRAD_POLITE_ROWS=0
run rows30000_$i P3D_WGRAD_POLITE_ROWS=30000
done
