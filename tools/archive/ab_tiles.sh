# A/B of the tile-order and tail switches of conv_igemm2.hip (tuning build only; the product build ignores the variables).
#   gpurun --timeout 900 -- 'bash tools/ab_tiles.sh'
P3D_EXTRA_HIPCC_FLAGS=-DP3D_TUNING python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "tuning build failed"; exit 1; }
mkdir -p gpurun_out/abt
run() { tag=$1; shift; env "$@" timeout -k 10 240 python bench.py $ARGS --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/abt/$tag.json 2> gpurun_out/abt/$tag.err; echo "$ARGS | $tag $(python3 -c "import json;print(json.load(open('gpurun_out/abt/$tag.json'))['ms_per_step'])")"; }
for ARGS in "" "--frames 32 --size 224"; do
for rep in 1 2; do
run xcd64_$rep P3D_TUNE_XCD_MIN_TILES=64
run xcd512_$rep P3D_TUNE_XCD_MIN_TILES=512
run xcdoff_$rep P3D_TUNE_XCD_MIN_TILES=1000000000
run notail_$rep P3D_TUNE_XCD_MIN_TILES=1000000000 P3D_TUNE_NO_TAIL=1
done
done
