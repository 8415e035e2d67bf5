#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for T in 0 1; do for S in 1 2 3 4 6 8 12 16; do
  export P3D_SPLITS=$S P3D_TILE=$T
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tune/t${T}_s${S} -- python3 tools/tune_igemm.py > /dev/null 2>&1
  f=$(ls gpurun_out/tune/t${T}_s${S}/*/*kernel_trace.csv | head -1)
  python3 - "$f" $T $S <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'igemm2' in r['Kernel_Name']]
d = [ (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
# 6 cases x 3 reps: take min of each triple
out = [min(d[i:i+3]) for i in range(0, len(d), 3)]
print('tile', sys.argv[2], 'splits', sys.argv[3], ' '.join('%6.1f' % v for v in out), flush=True)
PY
done; done
