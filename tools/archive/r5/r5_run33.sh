# what the side stream's LAUNCHES cost without their work: every filter gradient replaced by an empty kernel (WRONG results, timing only)
mkdir -p gpurun_out/r5ag
export P3D_LIB=$PWD/tools/ab/libp3dhip_wtune.so
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5ag/$tag.json 2> gpurun_out/r5ag/$tag.err; echo "$tag $(python3 -c "import json;d=json.loads(open('gpurun_out/r5ag/$tag.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"; }
for i in 1 2; do
run real_$i X=1
run empty1_$i P3D_TUNE_WGRAD_EMPTY=1
run empty2_$i P3D_TUNE_WGRAD_EMPTY=2
done
