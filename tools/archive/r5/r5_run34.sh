# one clean step of a kernel trace for profiles/r05_timeline.txt: the traced process's host falls behind on some steps (gaps on an idle GPU); keep the shortest
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5ah; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/trace.err
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
for i in 5 7 9 11 13 15; do python3 tools/timeline.py $f $i > $OUT/timeline_$i.txt; head -1 $OUT/timeline_$i.txt; done
rm -rf $OUT/trace
