#!/bin/bash
# Host cost of enqueueing one train step (890 launches) when N ranks share one host's cores -- rehearsal mode: every rank on GPU 0,
# no communicator.  The GPU box allows six processes on its card, and the launcher counts as one, so N = 1, 2, 4, 5.
#   gpurun -- 'tools/host_contention.sh gpurun_out/host'
out=${1:-gpurun_out/host}; mkdir -p $out
for n in 1 2 4 5; do
  P3D_BENCH_REHEARSAL=1 python bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline > $out/ranks_$n.json 2> $out/ranks_$n.err || { echo "ranks $n failed"; tail -3 $out/ranks_$n.err; }
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$out/ranks_$n.json") if l.startswith("{")][-1])
    print("ranks %d: host enqueue of one step into empty queues %.2f ms on rank 0, %.2f ms on the slowest rank (cores of this box: %s)" % ($n, d["host_enqueue_ms_one_step_empty_queue"], d["host_enqueue_ms_one_step_empty_queue_max_over_ranks"], __import__("os").cpu_count()))
except Exception as e:
    print("ranks $n: no line (%r)" % (e,))
PY
done
