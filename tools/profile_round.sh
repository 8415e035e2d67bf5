#!/bin/bash
# Profiles behind profiles/rNN_*: rocprofv3 kernel-trace stats of the default bench command, and the two PMC passes for
# HBM-side traffic (separate runs: --pmc must not be combined with other trace domains on this pool).
#   gpurun -- 'bash tools/profile_round.sh r02'
set -e
R=${1:-r02}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write.err
python3 tools/traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/traffic.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/stats
ls -la $OUT
