# The workloads beside the headline one (BASELINE configs[3] and [4]) -> gpurun_out/other/other_workloads.jsonl
#   gpurun --timeout 1100 -- 'bash tools/other_workloads.sh'
mkdir -p gpurun_out/other
out=gpurun_out/other/other_workloads.jsonl
: > $out
for st in concat unet++nonsa unet++ds gn_p3d gn_p3d_concat gn_p3d_decoder; do
  timeout -k 10 240 python bench.py --structure $st --steps 5 --warmup 2 --no-cpu-baseline >> $out 2> gpurun_out/other/$st.err || echo "{\"failed\": \"$st\"}" >> $out
  echo "$st done"
done
timeout -k 10 300 python bench.py --frames 32 --size 224 --steps 5 --warmup 2 --no-cpu-baseline >> $out 2> gpurun_out/other/big.err || echo '{"failed": "32x224x224"}' >> $out
timeout -k 10 300 python bench.py --frames 32 --size 224 --pointwise fp16 --steps 5 --warmup 2 --no-cpu-baseline >> $out 2> gpurun_out/other/big16.err || echo '{"failed": "32x224x224 fp16"}' >> $out
python3 -c "
import json
for l in open('$out'):
    d=json.loads(l)
    print(d.get('ms_per_step'), d.get('value'), d.get('config',{}).get('workload','')[:70] if 'config' in d else d)
"
