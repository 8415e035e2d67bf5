# is the side stream's cost a per-CU sharing cost?  The main stream on the upper half of the CU mask, the side stream on the lower half (disjoint), on
# the same half, or everywhere -- the main stream's handicap is the same in all, only the overlap differs (tuning build of net.hip; only
# CONTIGUOUS masks take effect on this runtime: strided ones -- every 2nd / 8th bit -- change nothing, lowest-32 gives 40-55 ms)
mkdir -p gpurun_out/r5am
export P3D_LIB=$PWD/tools/ab/libp3dhip_tune.so
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r5am/$tag.json 2> gpurun_out/r5am/$tag.err; echo "$tag $(python3 -c "import json;d=json.loads(open('gpurun_out/r5am/$tag.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"; }
for i in 1 2; do
run main_h128_side_all_$i P3D_MAIN_CU_MASK=h128
run main_h128_side_l128_$i P3D_MAIN_CU_MASK=h128 P3D_SIDE_CU_MASK=l128
run main_h128_side_h128_$i P3D_MAIN_CU_MASK=h128 P3D_SIDE_CU_MASK=h128
run main_h128_noside_$i P3D_MAIN_CU_MASK=h128 P3D_TUNE_SKIP_SIDE=deconv,results,block,stem
run main_l192_side_h192_$i P3D_MAIN_CU_MASK=l192 P3D_SIDE_CU_MASK=h192
done
