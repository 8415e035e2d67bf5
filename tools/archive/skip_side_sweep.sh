# The switches below exist only in a -DP3D_TUNING build of the library (the product build ignores them): rebuild on the GPU box first.
P3D_EXTRA_HIPCC_FLAGS=-DP3D_TUNING python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "tuning build failed"; exit 1; }
mkdir -p gpurun_out/r2s
S3=$(python3 -c "print(','.join('block%d/'%i for i in range(11,47)))")
S2=$(python3 -c "print(','.join('block%d/'%i for i in range(3,11)))")
S1=$(python3 -c "print(','.join('block%d/'%i for i in range(0,3)))")
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s/$tag.json 2> gpurun_out/r2s/$tag.err; echo "$tag $(python3 -c "import json;print(json.load(open('gpurun_out/r2s/$tag.json'))['ms_per_step'])")"; }
run base X=1
run dec P3D_TUNE_SKIP_SIDE=deconv,results
run s3 P3D_TUNE_SKIP_SIDE=$S3
run s2 P3D_TUNE_SKIP_SIDE=$S2
run s1 P3D_TUNE_SKIP_SIDE=$S1
run stem P3D_TUNE_SKIP_SIDE=stem
run all P3D_TUNE_SKIP_SIDE=deconv,results,block,stem
run base2 X=1
