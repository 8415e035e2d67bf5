#!/bin/bash
# A/B of the stem's one-pass filter gradient with the BatchNorm backward apply on its operand path (P3D_STEM_ONEPASS=0: the
# generic filter-gradient kernel, four passes, behind a separate apply launch).  Tuning build: the switches are compiled out
# of the product library.
# whatever this script builds into the package directory, the PRODUCT build is back when it exits (build.py also keys its
# object cache by the compile flags, so a later plain build would rebuild anyway)
trap 'env -u P3D_EXTRA_HIPCC_FLAGS python -c "
import sys; sys.path.insert(0, \".\")
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1' EXIT
P3D_EXTRA_HIPCC_FLAGS=-DP3D_TUNING python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "tuning build failed"; exit 1; }
set -e
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for v in 0 1; do
    P3D_STEM_ONEPASS=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline "$@" > gpurun_out/ab/stem1p_${v}_${i}.json 2>gpurun_out/ab/stem1p_${v}_${i}.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/stem1p_${v}_${i}.json").read().strip().splitlines()[-1])
print("one-pass stem=$v run $i: %.3f ms/step  %.1f clips/s" % (d["ms_per_step"], d["value"]))
PY
  done
done
