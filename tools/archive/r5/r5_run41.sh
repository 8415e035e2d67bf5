# filter gradients at full residency on a CONTIGUOUS slice of the CU mask, main stream unmasked (tuning build of net.hip + conv_wgrad2.hip)
mkdir -p gpurun_out/r5ao
export P3D_LIB=$PWD/tools/ab/libp3dhip_tune2.so
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5ao/$tag.json 2> gpurun_out/r5ao/$tag.err; echo "$tag $(python3 -c "import json;d=json.loads(open('gpurun_out/r5ao/$tag.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"; }
run base_1 X=1
run l64_kb48 P3D_SIDE_CU_MASK=l64 P3D_WGRAD_LDS_KB=48
run l96_kb48 P3D_SIDE_CU_MASK=l96 P3D_WGRAD_LDS_KB=48
run l128_kb48 P3D_SIDE_CU_MASK=l128 P3D_WGRAD_LDS_KB=48
run l160_kb48 P3D_SIDE_CU_MASK=l160 P3D_WGRAD_LDS_KB=48
run l192_kb48 P3D_SIDE_CU_MASK=l192 P3D_WGRAD_LDS_KB=48
run l128_kb55 P3D_SIDE_CU_MASK=l128 P3D_WGRAD_LDS_KB=55
run l192_kb55 P3D_SIDE_CU_MASK=l192 P3D_WGRAD_LDS_KB=55
run l192_kb82 P3D_SIDE_CU_MASK=l192
run l224_kb82 P3D_SIDE_CU_MASK=l224
run base_2 X=1
