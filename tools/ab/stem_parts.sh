#!/bin/bash
# where the stem kernel's time goes: builds with one ingredient removed (results are wrong in those builds; timing only)
mkdir -p gpurun_out/ab
for v in none "SW_EXP_NO_B -DSW_EXP_NO_A -DSW_EXP_NO_STAGE -DSW_EXP_NO_MATH" "SW_EXP_NO_B -DSW_EXP_NO_A -DSW_EXP_NO_STAGE -DSW_EXP_NO_MATH -DSW_EXP_NO_EPILOGUE" "SW_EXP_NO_B -DSW_EXP_NO_A -DSW_EXP_NO_STAGE -DSW_EXP_NO_MATH -DSW_EXP_NO_LOOP" "SW_EXP_NO_LOOP" "SW_EXP_NO_EPILOGUE"; do
  if [ "$v" = none ]; then F=""; else F="-D$v"; fi
  touch sap3d_tensorflow_amd/csrc/stem_wgrad.hip
  P3D_EXTRA_HIPCC_FLAGS="$F" python -c "
import sys; sys.path.insert(0,'.')
from sap3d_tensorflow_amd import build; build.build(force=False)" > /dev/null 2>gpurun_out/ab/parts_build.err || { echo "build failed"; tail -5 gpurun_out/ab/parts_build.err; exit 1; }
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernels > /dev/null 2> gpurun_out/ab/parts.err
  echo "$v: $(grep stem_wgrad_kernel gpurun_out/ab/parts.err)"
done
