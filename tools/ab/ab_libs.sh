#!/bin/bash
# tools/ab/ab_libs.sh <out dir> <rounds> "<lib names>" [bench.py args...]: same-box A/B of builds of the library (tools/ab/libp3dhip_<name>.so,
# `product` = the package's own), alternating, one JSON line per run kept under <out dir>.
out=$1; rounds=$2; libs=$3; shift 3
mkdir -p $out
for i in $(seq 1 $rounds); do
  for l in $libs; do
    if [ $l = product ]; then unset P3D_LIB; else export P3D_LIB=$PWD/tools/ab/libp3dhip_$l.so; fi
    python bench.py --no-cpu-baseline "$@" > $out/${l}_$i.json 2> $out/${l}_$i.err || { echo "$l run $i FAILED"; tail -3 $out/${l}_$i.err; continue; }
    python -c "
import json;d=json.loads(open('$out/${l}_$i.json').read().strip().splitlines()[-1]);r=d['roofline'];print('%-10s run $i  %8.3f ms/step %8.2f clips/s   %s %.2f us frac %.3f' % ('$l', d['ms_per_step'], d['value'], r['kernel'], r['avg_launch_us'], r['frac']))"
  done
done
unset P3D_LIB
