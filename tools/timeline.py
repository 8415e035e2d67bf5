"""Timeline of ONE train step from a rocprofv3 kernel trace: per hardware queue busy time and gaps, the largest gaps of the main
queue with what the side queue ran meanwhile, kernel totals per queue.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline
  python tools/timeline.py gpurun_out/trace/<host>/<pid>_kernel_trace.csv[.gz] [step index, default 8] > profiles/r03_timeline.txt
"""
import collections
import csv
import gzip
import io
import re
import sys


def name(r):
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*', '', n)[:46]


def main():
    path = sys.argv[1]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    f = io.TextIOWrapper(gzip.open(path)) if path.endswith('.gz') else open(path)
    rows = list(csv.DictReader(f))
    for r in rows:
        r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['q'] = r['Queue_Id']
    # one period of the steady state: from the first kernel of step k's forward (the stem's padding pass) to the first kernel of
    # step k + 1's -- the optimiser no longer marks a step boundary (its ranges run inside the backward pass)
    rows.sort(key=lambda r: r['s'])
    starts = [i for i, r in enumerate(rows) if 'stem_pad_kernel' in r['Kernel_Name']]
    loss = [i for i, r in enumerate(rows) if 'smooth_l1' in r['Kernel_Name']]
    step = rows[starts[k]:starts[k + 1]]
    loss = [i - starts[k] for i in loss if starts[k] <= i < starts[k + 1]] or [0]
    rows = step
    k = 0
    t0 = min(r['s'] for r in step); t1 = max(r['e'] for r in step)
    print("step %d of the trace: %.3f ms, %d dispatches; loss kernel at %.3f ms" % (k, (t1 - t0) / 1e6, len(step), (rows[loss[0]]['s'] - t0) / 1e6))
    byq = collections.defaultdict(list)
    for r in step:
        byq[r['q']].append(r)
    mq = max(byq, key=lambda q: len(byq[q]))
    for q, l in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        l.sort(key=lambda r: r['s'])
        busy = sum(r['e'] - r['s'] for r in l)
        gaps = sum(max(0, l[i + 1]['s'] - l[i]['e']) for i in range(len(l) - 1))
        print("queue %s%s: %4d dispatches, busy %.3f ms, gaps %.3f ms, first start %.3f ms, last end %.3f ms" %
              (q, " (main)" if q == mq else "", len(l), busy / 1e6, gaps / 1e6, (l[0]['s'] - t0) / 1e6, (l[-1]['e'] - t0) / 1e6))
    main_q = byq[mq]
    side = [r for r in step if r['q'] != mq]
    g = sorted(((main_q[i + 1]['s'] - main_q[i]['e'], i) for i in range(len(main_q) - 1)), reverse=True)
    print("main-queue gaps: %d of %d above 5 us, together %.3f ms" % (sum(1 for x, _ in g if x > 5000), len(g), sum(x for x, _ in g if x > 5000) / 1e6))
    for x, i in g[:25]:
        a, b = main_q[i]['e'], main_q[i + 1]['s']
        sd = sorted(set(name(s) for s in side if s['s'] < b and s['e'] > a))
        print("  %6.1f us at %7.3f ms  %-46s -> %-46s | other queues: %s" % (x / 1e3, (a - t0) / 1e6, name(main_q[i]), name(main_q[i + 1]), ", ".join(sd) or "-"))
    for q, l in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        agg = collections.defaultdict(lambda: [0, 0])
        for r in l:
            a = agg[name(r)]; a[0] += 1; a[1] += r['e'] - r['s']
        print("queue %s kernels:" % q)
        for kname, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
            print("   %-46s n=%4d %8.3f ms  avg %7.2f us" % (kname, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))


if __name__ == "__main__":
    main()
