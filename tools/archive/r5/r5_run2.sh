set -o pipefail
mkdir -p gpurun_out/r5b
for v in base sleep1 sleep0 plainst nodrain; do
  for cfg in "5 32 0" "5 32 1"; do
    echo "== gated_chain_$v $cfg" >> gpurun_out/r5b/gated_variants.log
    timeout -k 10 120 tools/micro/bin/gated_chain_$v $cfg 2>&1 | grep -v "^chain:" >> gpurun_out/r5b/gated_variants.log || { echo "failed" >> gpurun_out/r5b/gated_variants.log; }
  done
done
cat gpurun_out/r5b/gated_variants.log
for i in 1 2 3; do
  for l in r04 cur; do
    if [ $l = r04 ]; then export P3D_LIB=$PWD/tools/ab/libp3dhip_r04.so; else unset P3D_LIB; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r5b/ab_${l}_$i.json 2> gpurun_out/r5b/ab_${l}_$i.err
    python -c "
import json;d=json.loads(open('gpurun_out/r5b/ab_${l}_$i.json').read().strip().splitlines()[-1]);print('$l run $i', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
  done
done
unset P3D_LIB
P3D_MEASURE_GATES=gpurun_out/r5b/gates.json timeout -k 10 900 python -m pytest tests/test_gpu_full.py -m gpu -x -q -k "gn_head" > gpurun_out/r5b/gn_head_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r5b/gn_head_tests.log
