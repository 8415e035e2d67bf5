mkdir -p gpurun_out/r5n
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r5n/all_tests.log 2>&1; echo "all tests rc=$?"; tail -5 gpurun_out/r5n/all_tests.log
python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r5n/bench.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r5n/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['launches_per_step'])"
python -c "import __graft_entry__ as g; g.smoke()"
