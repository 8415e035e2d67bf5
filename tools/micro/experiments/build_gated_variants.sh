#!/bin/bash
# builds tools/micro/bin/gated_chain_<variant> from the product sources with one ingredient of the chain hand-off changed
# (objects go to /tmp: the product library is not touched)
set -e
C=sap3d_tensorflow_amd/csrc
mkdir -p tools/micro/bin
for v in "base:" "sleep1:-DP3D_CHAIN_SLEEP=1" "sleep0:-DP3D_CHAIN_SLEEP=0" "plainst:-DP3D_CHAIN_EXP_PLAIN_STORES" "nodrain:-DP3D_CHAIN_EXP_NO_DRAIN"; do
  name=${v%%:*}; flags=${v#*:}
  O=/tmp/gc_$name; mkdir -p $O
  for f in conv_igemm2 bn_small conv_wgrad2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags -c $C/$f.hip -o $O/$f.o & done
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags -c tools/micro/gated_chain.hip -o $O/gated_chain.o &
  wait
  hipcc --offload-arch=gfx950 $O/gated_chain.o $O/conv_igemm2.o $O/bn_small.o $O/conv_wgrad2.o -o tools/micro/bin/gated_chain_$name
done
