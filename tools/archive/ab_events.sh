# A/B of the cross-stream event flags (tuning build): P3D_TUNE_EVENT_SYSFENCE=1 restores the system-scope fence of the fork events.
P3D_EXTRA_HIPCC_FLAGS=-DP3D_TUNING python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "tuning build failed"; exit 1; }
mkdir -p gpurun_out/abe
run() { tag=$1; shift; env "$@" timeout -k 10 240 python bench.py $ARGS --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/abe/$tag.json 2> gpurun_out/abe/$tag.err; echo "$ARGS | $tag $(python3 -c "import json;print(json.load(open('gpurun_out/abe/$tag.json'))['ms_per_step'])")"; }
for ARGS in "" "--structure gn_p3d --steps 5"; do
for rep in 1 2; do
run nofence_$rep P3D_TUNE_EVENT_SYSFENCE=0
run sysfence_$rep P3D_TUNE_EVENT_SYSFENCE=1
done
done
