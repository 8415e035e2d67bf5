#!/bin/bash
# A/B of the LDS-DMA ring depth of the 64-row igemm tiles (P3D_RING64: 3 stages = 48.5 KB per block, 4 = 64.5 KB, 5 = 80.5 KB)
# whatever this script builds into the package directory, the PRODUCT build is back when it exits (build.py also keys its
# object cache by the compile flags, so a later plain build would rebuild anyway)
trap 'env -u P3D_EXTRA_HIPCC_FLAGS python -c "
import sys; sys.path.insert(0, \".\")
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>&1' EXIT
build() { P3D_EXTRA_HIPCC_FLAGS="-DP3D_TUNING -DP3D_RING64=$1" python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=True)" > /dev/null 2>gpurun_out/ab/ring_build_$1.err || { echo "build $1 failed"; tail -5 gpurun_out/ab/ring_build_$1.err; exit 1; }; }
mkdir -p gpurun_out/ab
run() { tag=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1]);print('$tag', d['ms_per_step'], d['value'])"; }
for i in 1 2; do
  for r in 3 2 4 5; do
    build $r || exit 1
    run ring${r}_112_$i --steps 30 --warmup 8
    run ring${r}_112b_$i --steps 30 --warmup 8
    if [ $r = 5 ]; then P3D_WGRAD_LDS_KB=76 python bench.py --no-cpu-baseline --steps 30 --warmup 8 | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('ring5 wgrad76KB', d['ms_per_step'])"; fi
    if [ $i = 1 ]; then run ring${r}_224 --frames 32 --size 224 --steps 4 --warmup 2; timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -1; fi
  done
done
