"""Data-parallel path on CPU with world_size 2 (gloo): the control plane bench.py / the drivers use, and the
arithmetic contract of the in-library all-reduce (SUM of per-shard gradients, because the reference loss is a
batch SUM, utils/network.py:60) checked with the oracle."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import p3d
    from sap3d_tensorflow_amd.dp import Plane
    plane = Plane()
    ident = plane.share_from_rank0(lambda: b"id-from-rank0" + bytes(115))
    lo, hi = plane.shard(4)
    # per-shard oracle gradients, summed over ranks == what every replica holds after the all-reduce
    cfg = p3d.NetConfig(base=8, blocks=(1, 1, 1))
    params = p3d.init_params(1, 'unet', cfg, dtype=np.float64)
    x = p3d.synthetic_clip(0, (4, 16, 32, 32, 3)).astype(np.float64)
    y = p3d.synthetic_target(3, (4, 16, 32, 32)).astype(np.float64)
    loss, _, grads, _ = p3d.loss_and_grads(params, x[lo:hi], y[lo:hi], 0.0, True, 'unet', cfg, np.float64)
    names = sorted(grads)
    arrs = [np.ascontiguousarray(grads[n]) for n in names]
    plane.sum_arrays(arrs)
    tmax = plane.max_over_ranks(1.0 + rank)
    plane.barrier()
    q.put((rank, ident[:13], (lo, hi), float(loss), {n: a for n, a in zip(names, arrs)}, tmax))
    plane.close()


def test_two_rank_control_plane_and_gradient_sum():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, id0, sh0, l0, g0, t0), (r1, id1, sh1, l1, g1, t1) = out
    assert id0 == id1 == b"id-from-rank0"
    assert sh0 == (0, 2) and sh1 == (2, 4)
    assert t0 == t1 == 2.0                      # MAX over ranks
    # both ranks hold the same summed gradients, equal to the sum of the two per-shard oracle gradients
    sys.path.insert(0, ROOT)
    from oracle import p3d
    cfg = p3d.NetConfig(base=8, blocks=(1, 1, 1))
    params = p3d.init_params(1, 'unet', cfg, dtype=np.float64)
    x = p3d.synthetic_clip(0, (4, 16, 32, 32, 3)).astype(np.float64)
    y = p3d.synthetic_target(3, (4, 16, 32, 32)).astype(np.float64)
    want = None
    for lo, hi in ((0, 2), (2, 4)):
        _, _, g, _ = p3d.loss_and_grads(params, x[lo:hi], y[lo:hi], 0.0, True, 'unet', cfg, np.float64)
        want = g if want is None else {n: want[n] + g[n] for n in g}
    for n in want:
        assert np.allclose(g0[n], want[n], rtol=1e-12, atol=1e-12), n
        assert np.array_equal(g0[n], g1[n]), n


def test_bare_bench_command_starts_the_ranks_as_a_child(monkeypatch):
    """`python bench.py --gpus N` with no torch.distributed.run environment must start the ranks itself, as a CHILD process
    (a process that has touched the GPU may never exec another program), pass the arguments through unchanged and hand back the
    child's exit code.  No GPU is needed to check the command it builds."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    try:
        bench.main()
        raise AssertionError("bench.main() must exit with the child's code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    # the rendezvous: torch.distributed.run's own free-port choice on the loopback address (a port picked here by bind-and-close
    # could be gone by the time the ranks bind it)
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")


def test_rank_to_device_under_both_launcher_conventions():
    """bench.py's device choice: LOCAL_RANK when every rank sees all the node's GPUs (torch.distributed.run), 0 when the launcher
    shows each rank exactly one; anything else cannot give one GPU per rank and must stop the run."""
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.pick_device(r, 8) for r in range(8)] == list(range(8))
    assert [bench.pick_device(r, 1) for r in range(8)] == [0] * 8
    assert bench.pick_device(2, -1) == 2
    try:
        bench.pick_device(3, 2)
        raise AssertionError("two visible GPUs cannot serve LOCAL_RANK 3")
    except SystemExit:
        pass


def test_rendezvous_keeps_standard_output_clean():
    """A bench run prints ONE JSON line on standard output; gloo's C++ side announces its connections on file descriptor 1
    ("[Gloo] Rank 0 is connected to 1 peer ranks"): Plane() sends whatever the rendezvous prints to standard error."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); from sap3d_tensorflow_amd.dp import Plane; "
            "p = Plane(); print('LINE', p.rank); p.close()" % ROOT)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port))) for r in range(2)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
        assert out == "LINE %d\n" % r, out
