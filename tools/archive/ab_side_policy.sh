# The switches below exist only in a -DP3D_TUNING build of the library (the product build ignores them): rebuild on the GPU box first.
P3D_EXTRA_HIPCC_FLAGS=-DP3D_TUNING python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "tuning build failed"; exit 1; }
mkdir -p gpurun_out/ab
run() { tag=$1; shift; env "$@" timeout -k 10 240 python bench.py $ARGS --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; echo "$ARGS | $tag $(python3 -c "import json;print(json.load(open('gpurun_out/ab/$tag.json'))['ms_per_step'])")"; }
for ARGS in "--structure unet++nonsa" "--frames 32 --size 224" "--structure gn_p3d"; do
run default X=1
run nodefer P3D_DEFER_SIDE=0
run lds48 P3D_WGRAD_LDS_KB=48
run both P3D_DEFER_SIDE=0 P3D_WGRAD_LDS_KB=48
done
