# what the optimiser costs the step: P3D_TUNE_SKIP_ADAM (tuning build of net.hip; the weights stay put, timing only)
mkdir -p gpurun_out/r5ar
export P3D_LIB=$PWD/tools/ab/libp3dhip_tune.so
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5ar/$tag.json 2> gpurun_out/r5ar/$tag.err; echo "$tag $(python3 -c "import json;d=json.loads(open('gpurun_out/r5ar/$tag.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"; }
for i in 1 2; do
run adam_$i X=1
run noadam_$i P3D_TUNE_SKIP_ADAM=1
done
