#!/bin/bash
# A/B: grouped filter-gradient launches on 64x128 tiles when every problem of the group has >= 2048 positions and 128-wide outputs
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
run() { tag=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1]);print('$tag', d['ms_per_step'], d['value'])"; }
for i in 1 2; do
  for v in 0 2048; do
    export P3D_TUNE_WGRAD_GROUP_RECT=$v
    run rect${v}_224_$i --frames 32 --size 224 --steps 5 --warmup 2
    run rect${v}_pp_$i --structure unet++nonsa --steps 5 --warmup 2
    run rect${v}_ds_$i --structure unet++ds --steps 5 --warmup 2
    run rect${v}_cat_$i --structure concat --steps 8 --warmup 2
    run rect${v}_gnd_$i --structure gn_p3d_decoder --steps 5 --warmup 2
    run rect${v}_112_$i --steps 30 --warmup 8
    run rect${v}_gn_$i --structure gn_p3d --steps 5 --warmup 2
  done
done
