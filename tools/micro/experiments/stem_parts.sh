#!/bin/bash
# Where the stem filter-gradient kernel's time goes (round 4): builds of libp3dhip with ONE ingredient of stem_wgrad.hip removed
# (results are WRONG in those builds; timing only).  The switches are not in the product source: stem_parts.patch puts them into
# a scratch copy of csrc/, the scratch library is used through P3D_LIB, and the product library is never touched
# (the earlier form of this script left its last variant installed as the product library).
set -e
ROOT=$(cd "$(dirname "$0")/../../.." && pwd)
S=$(mktemp -d /tmp/stem_parts.XXXXXX)
trap 'rm -rf "$S"' EXIT
mkdir -p "$S/pkg/sap3d_tensorflow_amd" "$S/pkg/include"      # net.hip includes ../../include/p3d_hip.h
cp -r "$ROOT/sap3d_tensorflow_amd/csrc" "$S/pkg/sap3d_tensorflow_amd/csrc"; cp "$ROOT/include/p3d_hip.h" "$S/pkg/include/"
ln -s "$S/pkg/sap3d_tensorflow_amd/csrc" "$S/csrc"
patch -s "$S/csrc/stem_wgrad.hip" < "$ROOT/tools/micro/experiments/stem_parts.patch"
mkdir -p "$ROOT/gpurun_out/ab"
SRC="conv_igemm2 conv_wgrad2 stem_wgrad elementwise bn_small gn cbam head attention attention_flash metrics net"
mkdir -p "$S/o"
for f in $SRC; do [ $f = stem_wgrad ] && continue; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT" -c "$S/csrc/$f.hip" -o "$S/o/$f.o" & done; wait
for v in none "SW_EXP_NO_B -DSW_EXP_NO_A -DSW_EXP_NO_STAGE -DSW_EXP_NO_MATH" "SW_EXP_NO_B -DSW_EXP_NO_A -DSW_EXP_NO_STAGE -DSW_EXP_NO_MATH -DSW_EXP_NO_EPILOGUE" "SW_EXP_NO_B -DSW_EXP_NO_A -DSW_EXP_NO_STAGE -DSW_EXP_NO_MATH -DSW_EXP_NO_LOOP" "SW_EXP_NO_LOOP" "SW_EXP_NO_EPILOGUE"; do
  if [ "$v" = none ]; then F=""; else F="-D$v"; fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $F -c "$S/csrc/stem_wgrad.hip" -o "$S/o/stem_wgrad.o"
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$S/libp3dhip_parts.so" $(for f in $SRC; do echo "$S/o/$f.o"; done) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  (cd "$ROOT" && P3D_LIB="$S/libp3dhip_parts.so" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernels > /dev/null 2> gpurun_out/ab/parts.err)
  echo "$v: $(grep stem_wgrad_kernel "$ROOT/gpurun_out/ab/parts.err")"
done
