"""Compare the planner's choice with forced plans, per op: python tools/plan_sweep.py <dir> <igemm|wgrad> <min gain us> <plan tag> <forced tags...>
where <dir>/t<tag>.txt are outputs of `tools/op_times.py . --igemm-tile T` (or --wgrad-tile / --igemm-splits); prints the ops on
which a forced plan beats the planner by more than the threshold."""
import re,collections,sys
d=sys.argv[1]; kind=sys.argv[2]; thr=float(sys.argv[3])
T=sys.argv[4:]
tab=collections.OrderedDict()
for t in T:
    for l in open("%s/t%s.txt"%(d,t)):
        m=re.match(r"(\S+)\s+(\S+)\s+ph(\d)\s+([\d.]+) us",l)
        if not m or kind not in m.group(2): continue
        key=(m.group(1),m.group(3))
        tab.setdefault(key,{}).setdefault(t,[]).append((m.group(2),float(m.group(4))))
tot=0
for k,v in tab.items():
    base=sum(x[1] for x in v.get(T[0],[]))
    best=min((sum(x[1] for x in v.get(t,[])),t) for t in T[1:] if t in v)
    if base-best[0]>thr:
        tot+=base-best[0]
        row=" ".join("%s:%8.1f"%(t,sum(x[1] for x in v.get(t,[]))) for t in T[1:])
        print("%-26s ph%s plan %-28s %8.1f | %s <== %.1f"%(k[0],k[1],",".join(x[0] for x in v.get(T[0],[]))[:28],base,row,base-best[0]))
print("total possible", tot)
