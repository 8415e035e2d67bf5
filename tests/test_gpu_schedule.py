"""Scheduler-level test: the stream operations of ONE train step of p3d_unet -- launches on the main,
side (filter gradients) and comm (all-reduce) streams, async fills, event records / waits -- as p3d_debug_schedule reports them,
checked as a happens-before graph:

 * every wait names an event that was recorded before it in the same pass;
 * the gradient buffer's zero fill (side stream, beside the forward pass) precedes every launch of the backward pass, on
   every stream;
 * every all-reduce hand-over comes after every launch issued before it on the main and side streams (the gradients of its
   range are final), and Adam comes after every all-reduce;
 * the last optimiser launch comes after EVERY launch of the step, the first after everything but the stem's filter gradient
   (the designed overlap: that part updates every variable but the stem kernel);
 * at the end of the step the main stream has joined the side and comm streams, so whatever the next call enqueues is ordered
   behind all of it;
 * and the synchronisation skeleton (who records, who waits, in which order, around which runs of launches) equals the
   committed one (tests/golden/schedule_unet_*.txt) -- a change of the scheduler has to be looked at and re-recorded:
       P3D_WRITE_SCHEDULE_GOLDEN=gpurun_out/sched python -m pytest tests/test_gpu_schedule.py -m gpu

The ordering bugs this code can have (round 3: a fill that was not ordered against the launch behind it) do not show as wrong
numbers until another session reuses the state; they do show here."""
import os
import re

import numpy as np
import pytest

from oracle import p3d

HERE = os.path.dirname(os.path.abspath(__file__))
WRITE_TO = os.environ.get("P3D_WRITE_SCHEDULE_GOLDEN")


def parse(lines):
    ops = []
    for ln in lines:
        parts = ln.split()
        kind, stream = parts[0], parts[1]
        rest = " ".join(parts[2:])
        ops.append((kind, stream, rest))
    return ops


def happens_before(ops):
    """Vector clocks over the trace: clock[i][s] = number of operations of stream s that are ordered before operation i
    (operation i itself included on its own stream)."""
    streams = sorted({s for _, s, _ in ops})
    cur = {s: {t: 0 for t in streams} for s in streams}          # per stream: what it has seen of every stream
    at_record = {}
    out = []
    for kind, s, rest in ops:
        cur[s][s] += 1
        if kind == "R":
            at_record[rest] = dict(cur[s])
        elif kind == "W":
            assert rest in at_record, "stream %s waits for %s, which no operation of this pass has recorded" % (s, rest)
            for t, v in at_record[rest].items():
                cur[s][t] = max(cur[s][t], v)
        out.append(dict(cur[s]))
    return streams, out


def ordered(i, j, ops, clocks, pos):
    """operation i happens-before operation j"""
    si = ops[i][1]
    return clocks[j][si] >= pos[i]


def skeleton(ops):
    """The synchronisation skeleton: fills, records, waits and all-reduces as they are; every run of launches of one stream
    between them as one line with its length and its first and last kernel."""
    out, run = [], []

    def flush():
        if run:
            first, last = run[0], run[-1]
            out.append("L %s x%d  %s .. %s" % (first[1], len(run), first[2], last[2]))
            run.clear()
    for op in ops:
        if op[0] == "L":
            if run and run[0][1] != op[1]:
                flush()
            run.append(op)
        else:
            flush()
            rest = re.sub(r"allreduce \d+ \d+", "allreduce", op[2])
            out.append("%s %s %s" % (op[0], op[1], rest))
    flush()
    return out


def check_schedule(lines, with_comm):
    ops = parse(lines)
    streams, clocks = happens_before(ops)                       # (asserts every wait's event was recorded in this pass)
    pos, count = [], {s: 0 for s in streams}
    for _, s, _ in ops:
        count[s] += 1
        pos.append(count[s])
    launches = [i for i, op in enumerate(ops) if op[0] == "L"]
    assert "main" in streams and "side" in streams and len(launches) > 100
    # -- the gradient buffer is zeroed before anything of the backward pass runs, on whichever stream
    fills = [i for i, op in enumerate(ops) if op[0] == "M" and op[2] == "gradients"]
    assert len(fills) == 1
    backward = [i for i in launches if "@" in ops[i][2]]
    assert backward
    for i in backward:
        assert ordered(fills[0], i, ops, clocks, pos), ("launch not ordered behind the gradient fill", ops[i])
    # -- the optimiser
    adam = [i for i in launches if ops[i][2].startswith("adam_kernel")]
    assert len(adam) == 2 and all(ops[i][1] == "main" for i in adam)
    others = [i for i in launches if i not in adam]
    for i in others:
        assert ordered(i, adam[-1], ops, clocks, pos), ("the last optimiser launch does not wait for", ops[i])
        if not (ops[i][1] == "side" and ops[i][2].endswith("@stem/conv")):
            assert ordered(i, adam[0], ops, clocks, pos), ("the first optimiser launch does not wait for", ops[i])
    stem_side = [i for i in launches if ops[i][1] == "side" and ops[i][2].endswith("@stem/conv")]
    assert stem_side and not ordered(stem_side[0], adam[0], ops, clocks, pos), "the designed overlap is gone (or the trace lost its tags)"
    # -- all-reduce hand-overs
    comms = [i for i, op in enumerate(ops) if op[0] == "C"]
    assert bool(comms) == with_comm
    covered = 0
    for c in comms:
        lo, hi = [int(v) for v in ops[c][2].split()[1:3]]
        assert 0 <= lo < hi
        covered += hi - lo
        for i in launches:
            if i < c and ops[i][1] in ("main", "side") and i not in adam:
                assert ordered(i, c, ops, clocks, pos), ("all-reduce handed over before", ops[i])
        assert any(ordered(c, a, ops, clocks, pos) for a in adam), "no optimiser launch behind this all-reduce"
    if comms:
        los = sorted(int(ops[c][2].split()[1]) for c in comms)
        assert los[0] == 0                                      # the buckets reach the start of the buffer ...
        assert ordered(max(comms), adam[-1], ops, clocks, pos)
    # -- the step ends joined: the main stream has seen everything
    end = clocks[-1] if ops[-1][1] == "main" else None
    last_main = max(i for i, op in enumerate(ops) if op[1] == "main")
    for s in streams:
        assert clocks[last_main][s] == count[s], ("main stream ends the step without having joined", s, clocks[last_main][s], count[s])
    return skeleton(ops)


def compare_or_write(name, skel):
    path = os.path.join(HERE, "golden", name)
    if WRITE_TO:
        os.makedirs(WRITE_TO, exist_ok=True)
        with open(os.path.join(WRITE_TO, name), "w") as f:
            f.write("\n".join(skel) + "\n")
        return
    want = open(path).read().splitlines()
    assert skel == want, "\n".join(["the synchronisation skeleton of the step changed (tests/test_gpu_schedule.py):"] +
                                   [a if a == b else "  now: %s\n  was: %s" % (a, b) for a, b in zip(skel, want)][:40] +
                                   ["lengths %d / %d" % (len(skel), len(want))])


@pytest.mark.gpu
@pytest.mark.parametrize("with_comm", [False, True])
def test_train_step_schedule_of_unet(with_comm, monkeypatch):
    from sap3d_tensorflow_amd import P3DSession
    cfg, shape = p3d.NetConfig(base=16, blocks=(1, 1, 2)), (2, 16, 32, 32)
    if with_comm:
        monkeypatch.setenv("P3D_BUCKET_MB", "1")                # several buckets on this small graph
    s = P3DSession('unet', batch=shape[0], frames=shape[1], height=shape[2], width=shape[3], base=cfg.base, blocks=cfg.blocks, seed=1)
    if with_comm:
        s.comm_init(P3DSession.comm_unique_id())                # one rank: the schedule of the data-parallel step, sums of one
    s.upload(p3d.synthetic_clip(0, shape + (3,)), p3d.synthetic_target(3, shape))
    s.train_step_device(0.5, seed=0)                            # scratch buffers sized, plans made
    s.synchronize()
    first = s.schedule(0.5, seed=1)
    again = s.schedule(0.5, seed=2)
    assert first == again                                       # the launch list is static
    skel = check_schedule(first, with_comm)
    compare_or_write("schedule_unet_b16_112%s.txt" % ("_comm" if with_comm else ""), skel)
    assert np.isfinite(s.last_loss())
    s.close()


def test_happens_before_machinery_catches_a_missing_join():
    """CPU: the checker itself -- a trace whose optimiser does not wait for the side stream must fail."""
    good = ["R main e0", "W side e0", "M side gradients", "R side e1", "M main loss", "W main e1"] + \
           ["L main k%d" % i for i in range(100)] + ["L main bwd @op"] + \
           ["R main e2", "W side e2", "L side wgrad2 @op", "R main e3", "W side e3", "L side wgrad2 @stem/conv", "R side e4",
            "L main adam_kernel @x", "W main e4", "L main adam_kernel @x"]
    with pytest.raises(AssertionError, match="first optimiser launch does not wait"):
        check_schedule(good, False)          # (the first Adam part must at least wait for the non-stem filter gradient)
    fixed = good[:-6] + ["R side e5", "R main e3", "W side e3", "L side wgrad2 @stem/conv", "R side e4", "W main e5",
                         "L main adam_kernel @x", "W main e4", "L main adam_kernel @x"]
    check_schedule(fixed, False)
    unjoined = fixed[:-2] + ["L main adam_kernel @x"]
    with pytest.raises(AssertionError):
        check_schedule(unjoined, False)
    with pytest.raises(AssertionError, match="no operation of this pass has recorded"):
        check_schedule(["W main e9"] + fixed, False)
