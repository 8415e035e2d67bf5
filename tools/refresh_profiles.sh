#!/bin/bash
# Everything behind profiles/rNN_{bench_*,traffic,mfma_util,per_layer,timeline}: one gpurun call.
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh r04'
set -e
R=${1:-r04}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
bash tools/profile_round.sh $R > $OUT/profile_round.log 2>&1
python3 bench.py --steps 20 --warmup 5 --per-layer $OUT/per_layer.csv --dump-launches $OUT/launches.csv > $OUT/bench_line.json 2> $OUT/bench_line.err
tail -c 1500 $OUT/bench_line.json
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/trace.err
python3 tools/timeline.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 8 > $OUT/timeline.txt
rm -rf $OUT/trace
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
tail -2 $OUT/smoke.log
