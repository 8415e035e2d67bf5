# A/B of how much decoder filter-gradient work is parked until the backward walk reaches the encoder (tuning build).
P3D_EXTRA_HIPCC_FLAGS=-DP3D_TUNING python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "tuning build failed"; exit 1; }
mkdir -p gpurun_out/abd
run() { tag=$1; shift; env "$@" timeout -k 10 240 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/abd/$tag.json 2> gpurun_out/abd/$tag.err; echo "$tag $(python3 -c "import json;print(json.load(open('gpurun_out/abd/$tag.json'))['ms_per_step'])")"; }
for rep in 1 2; do
for b in 130 100 45 20 0; do
run budget${b}_$rep P3D_TUNE_DEFER_GFLOP=$b
done
done
