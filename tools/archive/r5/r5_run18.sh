# what the side stream's filter gradients cost the step today (timing diagnostic: P3D_TUNE_SKIP_SIDE drops them; tuning build of net.hip)
mkdir -p gpurun_out/r5r
export P3D_LIB=$PWD/tools/ab/libp3dhip_tune.so
S3=$(python3 -c "print(','.join('block%d/'%i for i in range(11,47)))")
S2=$(python3 -c "print(','.join('block%d/'%i for i in range(3,11)))")
S1=$(python3 -c "print(','.join('block%d/'%i for i in range(0,3)))")
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5r/$tag.json 2> gpurun_out/r5r/$tag.err; echo "$tag $(python3 -c "import json;print(json.loads(open('gpurun_out/r5r/$tag.json').read().strip().splitlines()[-1])['ms_per_step'])")"; }
run base X=1
run dec P3D_TUNE_SKIP_SIDE=deconv,results
run s3 P3D_TUNE_SKIP_SIDE=$S3
run s2 P3D_TUNE_SKIP_SIDE=$S2
run s1 P3D_TUNE_SKIP_SIDE=$S1
run stem P3D_TUNE_SKIP_SIDE=stem
run all P3D_TUNE_SKIP_SIDE=deconv,results,block,stem
run base2 X=1
