#!/bin/bash
# Profiles behind profiles/rNN_*: rocprofv3 kernel-trace stats of the default bench command, and the two PMC passes for
# HBM-side traffic (separate runs: --pmc must not be combined with other trace domains on this pool).
#   gpurun -- 'bash tools/profile_round.sh r03'
set -e
R=${1:-r03}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write.err
python3 tools/traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/traffic.json
# matrix-core utilisation and stall shares: one SQ pass (8 SQ slots) + GRBM
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/sq.err
python3 tools/mfma_util.py $OUT/pmc_sq $OUT/mfma_util.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/stats
ls -la $OUT
