#!/bin/bash
# tools/ab/build_variant.sh <name> <source.hip> [extra hipcc flags...]: links tools/ab/libp3dhip_<name>.so from the product's objects
# with ONE translation unit rebuilt from <source.hip> (a variant of a file of csrc/, compiled with the extra flags).  The product
# library is not touched; use the result through P3D_LIB.
set -e
name=$1; src=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/sap3d_tensorflow_amd/build
unit=$(basename "$src" .hip); unit=${unit%%__*}           # conv_igemm2__loader4.hip replaces conv_igemm2.o
O=/tmp/variant_$name; mkdir -p $O
V="$ROOT/sap3d_tensorflow_amd/csrc/.variant_${name}_$unit.hip"; cp "$src" "$V"
trap "rm -f $V" EXIT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value "$@" -c "$V" -o $O/$unit.o
objs=""
for f in conv_igemm2 conv_pointwise conv_wgrad2 stem_wgrad elementwise bn_small gn cbam head attention attention_flash metrics net; do
  if [ $f = $unit ]; then objs="$objs $O/$unit.o"; else objs="$objs $B/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/libp3dhip_$name.so" $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "$ROOT/tools/ab/libp3dhip_$name.so"
