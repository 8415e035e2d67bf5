# The switches below exist only in a -DP3D_TUNING build of the library (the product build ignores them): rebuild on the GPU box first.
P3D_EXTRA_HIPCC_FLAGS=-DP3D_TUNING python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "tuning build failed"; exit 1; }
# bench.py with the filter-gradient kernel's LDS request (= blocks per CU) and the plan's slots per CU swept
#   gpurun -- 'CASES="82:3 82:6" bash tools/wgrad_lds_sweep.sh'
mkdir -p gpurun_out/lds
for c in ${CASES:-48:3 55:3 82:3}; do
  kb=${c%%:*}; sl=${c##*:}
  P3D_WGRAD_LDS_KB=$kb P3D_WGRAD_SLOTS=$sl timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/lds/$kb_$sl.json 2> gpurun_out/lds/$kb_$sl.err
  echo "lds_kb=$kb slots=$sl $(python3 -c "import json;print(json.load(open('gpurun_out/lds/$kb_$sl.json'))['ms_per_step'])")"
done
