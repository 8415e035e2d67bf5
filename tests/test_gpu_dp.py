"""The data-parallel gradient hand-over of libp3dhip, as far as ONE GPU can exercise it (VERDICT / ADVICE round 1):

 * bucket audit for every structure: with small buckets, the ranges handed to the all-reduce must tile the flat
   gradient buffer [0, n_train) exactly once, back to front, and every element must already hold its FINAL value when
   its bucket is handed over (a bucket launched before its last producer had run would silently drop gradient terms);
 * a one-rank RCCL communicator reduces to the identity: with it, the trajectory must equal the one without, bit for bit
   (bucketed launches on the comm stream, the joins with the main and side streams, torch loaded in the same process).

No scaling curve can be measured here; multi-rank arithmetic is covered on CPU by tests/test_dp_gloo.py."""
import os

import numpy as np
import pytest

from oracle import p3d

pytestmark = pytest.mark.gpu

AUDIT = [
    ("unet", p3d.NetConfig(base=16, blocks=(2, 2, 3)), (2, 16, 48, 48)),
    ("concat", p3d.NetConfig(base=16, blocks=(1, 2, 2)), (2, 16, 32, 32)),
    ("unet++nonsa", p3d.NetConfig(base=16, blocks=(1, 1, 2)), (2, 16, 32, 32)),      # ops run out of variable-creation order
    ("unet++ds", p3d.NetConfig(base=16, blocks=(1, 1, 2)), (2, 16, 32, 32)),
    ("gn_p3d", p3d.NetConfig(base=16, blocks=(1, 2, 2)), (2, 16, 32, 32)),
    ("gn_p3d_concat", p3d.NetConfig(base=16, blocks=(1, 1, 2)), (2, 16, 32, 32)),
    ("gn_p3d_decoder", p3d.NetConfig(base=16, blocks=(1, 1, 2)), (2, 16, 32, 32)),
]


@pytest.mark.parametrize("structure,cfg,shape", AUDIT)
@pytest.mark.parametrize("bucket_floats", [1 << 12, 1 << 16])
def test_buckets_tile_the_gradient_buffer_and_are_final(structure, cfg, shape, bucket_floats):
    from sap3d_tensorflow_amd import P3DSession
    s = P3DSession(structure, batch=shape[0], frames=shape[1], height=shape[2], width=shape[3], base=cfg.base,
                   blocks=cfg.blocks, seed=1)
    s.upload(p3d.synthetic_clip(0, shape + (3,)), p3d.synthetic_target(3, shape))
    buckets, n_train, stale = s.bucket_audit(bucket_floats, dropout=0.5, seed=3)
    assert stale == 0, "%d gradient elements were handed to the all-reduce before their last producer ran" % stale
    assert len(buckets) >= 2
    # back to front, contiguous, exactly once
    hi_expected = n_train
    for lo, hi, after_op in buckets:
        assert hi == hi_expected and 0 <= lo < hi, (lo, hi, hi_expected)
        hi_expected = lo
    assert hi_expected == 0
    # every bucket collects at least the requested size, except the two that end the pass: what is final before the first
    # op's filter gradient is handed over ahead of it (so neither its all-reduce nor its Adam waits for that launch), and
    # the first op's own variables after it; ops are walked in reverse
    assert all(hi - lo >= bucket_floats for lo, hi, _ in buckets[:-2])
    ops = [op for _, _, op in buckets]
    assert ops == sorted(ops, reverse=True)
    # the audit's gradients are the ordinary ones (another bucket size regroups the filter-gradient launches, which may
    # change the last bit of a K-cut sum: compare to fp32 rounding, not bitwise)
    names = [n for n, _, tr in s.variables() if tr]
    g_audit = {n: s.get_grad(n) for n in names[:8]}
    s.backward(p3d.synthetic_clip(0, shape + (3,)), p3d.synthetic_target(3, shape), 0.5, seed=3)
    for n, g in g_audit.items():
        assert np.allclose(g, s.get_grad(n), rtol=1e-4, atol=1e-6 * max(np.abs(g).max(), 1e-30)), n
    s.close()


# every structure with many small buckets, and the headline one with ONE bucket that swallows the whole buffer: then the
# parked decoder jobs, the two-part optimiser step and the single hand-over all coincide at the last boundary
IDENTITY = [(st, "1") for st in ("unet", "concat", "unet++nonsa", "unet++ds", "gn_p3d", "gn_p3d_concat", "gn_p3d_decoder")] + [("unet", "4096")]


@pytest.mark.parametrize("structure,bucket_mb", IDENTITY)
def test_single_rank_allreduce_is_identity(structure, bucket_mb, monkeypatch):
    import torch  # noqa: F401  (same process as libp3dhip, like bench.py at N > 1)
    from sap3d_tensorflow_amd import P3DSession
    cfg = p3d.NetConfig(base=16, blocks=(2, 2, 3)) if structure == "unet" else p3d.NetConfig(base=16, blocks=(1, 1, 2))
    shape = (2, 16, 48, 48) if structure == "unet" else (2, 16, 32, 32)
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)

    # read at create (the bucket walk also fixes where the queued filter gradients are flushed, so both runs must share it)
    monkeypatch.setenv("P3D_BUCKET_MB", bucket_mb)

    def run(with_comm):
        s = P3DSession(structure, batch=shape[0], frames=shape[1], height=shape[2], width=shape[3], base=cfg.base,
                       blocks=cfg.blocks, seed=2)
        assert s.comm_info()[0] == 0                     # no communicator yet
        if with_comm:
            s.comm_init(P3DSession.comm_unique_id())
            assert s.comm_info() == (1, 0, 0)            # what RCCL says: one rank, rank 0, device 0 (bench.py's "rccl_ranks")
        losses = [np.float32(s.train_step(x, y, dropout=0.0)) for _ in range(3)]
        w = {n: s.get_param(n) for n, _, _ in s.variables()}
        s.close()
        return losses, w

    l0, w0 = run(False)
    l1, w1 = run(True)
    # a one-rank all-reduce is the identity and every sum of the step is order-fixed: exact equality
    assert [a.tobytes() for a in l0] == [b.tobytes() for b in l1]
    for n in w0:
        assert np.array_equal(w0[n], w1[n]), n


def test_bench_launch_path_with_two_ranks_on_one_gpu(tmp_path):
    """The driver's multi-GPU command line in its BARE form (`python bench.py --gpus 2 ...`, no torch.distributed.run around
    it: bench.py starts the ranks itself as a child process and relays rank 0's ONE JSON line and the exit code), rehearsed on
    a one-GPU box: P3D_BENCH_REHEARSAL=1 puts both ranks on device 0 and builds no RCCL communicator (RCCL refuses two ranks
    on one device), so this covers the launch, the rendezvous, the barriers, the max-over-ranks timing and the output
    contract -- not the all-reduce."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["P3D_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    assert "rehearsal" in out and out["value"] > 0 and np.isfinite(out["final_loss"])
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert key in out
