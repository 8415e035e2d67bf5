set -o pipefail
mkdir -p gpurun_out/r5c
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r5c/ops_tests.log 2>&1; echo "ops tests rc=$?"; tail -5 gpurun_out/r5c/ops_tests.log
for i in 1 2 3; do
  for l in r04 cur; do
    if [ $l = r04 ]; then export P3D_LIB=$PWD/tools/ab/libp3dhip_r04.so; else unset P3D_LIB; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r5c/ab_${l}_$i.json 2> gpurun_out/r5c/ab_${l}_$i.err
    python -c "
import json;d=json.loads(open('gpurun_out/r5c/ab_${l}_$i.json').read().strip().splitlines()[-1]);print('$l run $i', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
  done
done
unset P3D_LIB
timeout -k 10 300 tools/micro/bin/conv_chain 3 quick > gpurun_out/r5c/conv_chain.log 2>&1; echo "conv_chain rc=$?"; grep -c NONDET gpurun_out/r5c/conv_chain.log
for l in r04 cur; do
  if [ $l = r04 ]; then export P3D_LIB=$PWD/tools/ab/libp3dhip_r04.so; else unset P3D_LIB; fi
  python bench.py --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r5c/b32_$l.json 2>/dev/null
  python bench.py --frames 32 --size 224 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r5c/s224_$l.json 2>/dev/null
  python -c "
import json
for t in ('b32','s224'):
    d=json.loads(open('gpurun_out/r5c/%s_$l.json'%t).read().strip().splitlines()[-1]);print('$l', t, d['ms_per_step'], d['value'])"
done
