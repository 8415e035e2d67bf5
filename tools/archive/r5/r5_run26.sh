# filter gradients launched per problem / per pair beside their own layer's input gradient instead of in groups of six (tuning build of net.hip + conv_wgrad2.hip)
mkdir -p gpurun_out/r5z
export P3D_LIB=$PWD/tools/ab/libp3dhip_tune2.so
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r5z/$tag.json 2> gpurun_out/r5z/$tag.err; echo "$tag $(python3 -c "import json;d=json.loads(open('gpurun_out/r5z/$tag.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['launches_per_step'])")"; }
run base_1 X=1
run g1_kb82 P3D_WGRAD_GROUP_MAX=1
run g1_kb48 P3D_WGRAD_GROUP_MAX=1 P3D_WGRAD_LDS_KB=48
run g1_kb55 P3D_WGRAD_GROUP_MAX=1 P3D_WGRAD_LDS_KB=55
run g2_kb82 P3D_WGRAD_GROUP_MAX=2
run g2_kb48 P3D_WGRAD_GROUP_MAX=2 P3D_WGRAD_LDS_KB=48
run g4_kb82 P3D_WGRAD_GROUP_MAX=4
run base_2 X=1
