# A/B on ONE box: 128x128 filter-gradient tiles for single launches with 128 or more of them (P3D_TUNE_WGRAD_BIG=1) against 64x128.
P3D_EXTRA_HIPCC_FLAGS=-DP3D_TUNING python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "tuning build failed"; exit 1; }
mkdir -p gpurun_out/abw
run() { tag=$1; shift; env "$@" timeout -k 10 240 python bench.py $ARGS --no-cpu-baseline > gpurun_out/abw/$tag.json 2> gpurun_out/abw/$tag.err; echo "$ARGS | $tag $(python3 -c "import json;print(json.load(open('gpurun_out/abw/$tag.json'))['ms_per_step'])")"; }
for ARGS in "--steps 30 --warmup 5" "--structure gn_p3d --steps 8 --warmup 2" "--structure unet++nonsa --steps 8 --warmup 2" "--structure gn_p3d_decoder --steps 8 --warmup 2" "--frames 32 --size 224 --steps 8 --warmup 2"; do
for rep in 1 2; do
run r64x128_$rep P3D_TUNE_WGRAD_BIG=0
run r128x128_$rep P3D_TUNE_WGRAD_BIG=1
done
done
