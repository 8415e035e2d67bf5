"""Bit-reproducibility of the HIP path: K-sliced convolutions, weight gradients, BatchNorm /
GroupNorm statistics and every other cross-block sum are folded in a fixed order (no floating-point atomics), so two
runs on the same inputs must agree bit for bit -- gradients, loss, and whole Adam trajectories."""
import numpy as np
import pytest

from oracle import p3d

pytestmark = pytest.mark.gpu

CASES = [
    ("unet", p3d.NetConfig(base=16, blocks=(2, 2, 3)), (2, 16, 48, 48)),
    ("unet", p3d.NetConfig(base=32, blocks=(1, 1, 2)), (8, 16, 32, 32)),
    ("concat", p3d.NetConfig(base=16, blocks=(1, 2, 2)), (2, 16, 32, 32)),
    ("gn_p3d", p3d.NetConfig(base=16, blocks=(1, 2, 2)), (2, 16, 32, 32)),
    ("gn_p3d_decoder", p3d.NetConfig(base=16, blocks=(1, 1, 2)), (2, 16, 32, 32)),
    ("unet++nonsa", p3d.NetConfig(base=16, blocks=(1, 1, 2)), (2, 16, 32, 32)),
    ("unet++ds", p3d.NetConfig(base=16, blocks=(1, 1, 2)), (2, 16, 32, 32)),
]


def _session(structure, cfg, shape, seed=1):
    from sap3d_tensorflow_amd import P3DSession
    return P3DSession(structure, batch=shape[0], frames=shape[1], height=shape[2], width=shape[3], base=cfg.base,
                      blocks=cfg.blocks, seed=seed)


@pytest.mark.parametrize("structure,cfg,shape", CASES)
def test_two_backward_calls_are_bit_identical(structure, cfg, shape):
    s = _session(structure, cfg, shape)
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    runs = []
    for _ in range(3):
        loss, pred = s.backward(x, y, 0.5, seed=11)          # dropout on: the mask is a pure function of the seed
        runs.append((np.float32(loss), pred.copy(), {n: s.get_grad(n) for n, _, tr in s.variables() if tr}))
    for other in runs[1:]:
        assert runs[0][0].tobytes() == other[0].tobytes()
        assert np.array_equal(runs[0][1], other[1])
        for n, g in runs[0][2].items():
            assert np.array_equal(g, other[2][n]), n
    assert all(np.isfinite(g).all() for g in runs[0][2].values())
    s.close()


@pytest.mark.parametrize("structure,cfg,shape", CASES[:4])
def test_adam_trajectories_are_bit_identical(structure, cfg, shape):
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    finals = []
    for _ in range(2):
        s = _session(structure, cfg, shape, seed=5)
        losses = [np.float32(s.train_step(x, y, dropout=0.5, seed=100 + i)) for i in range(3)]
        finals.append((losses, {n: s.get_param(n) for n, _, _ in s.variables()}))
        s.close()
    assert [l.tobytes() for l in finals[0][0]] == [l.tobytes() for l in finals[1][0]]
    for n, v in finals[0][1].items():
        assert np.array_equal(v, finals[1][1][n]), n


def test_captured_step_graph_replays_the_eager_step_bit_for_bit():
    """P3D_GRAPH=1 (opt-in): the train step captured into a hipGraph over three streams -- parked side-stream jobs, the
    two-part optimiser step and the device-resident dropout seed / Adam step size included -- must give the eager
    trajectory bit for bit.  The switch is read once per process, so both runs are child processes."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import p3d\n"
        "from sap3d_tensorflow_amd import P3DSession\n"
        "shape = (2, 16, 48, 48)\n"
        "s = P3DSession('unet', batch=2, frames=16, height=48, width=48, base=16, blocks=(2, 2, 3), seed=3)\n"
        "s.set_adam(1e-3)\n"
        "s.upload(p3d.synthetic_clip(0, shape + (3,)), p3d.synthetic_target(3, shape))\n"
        "losses = []\n"
        "for i in range(4):\n"
        "    s.train_step_device(0.5, seed=50 + i)\n"
        "    losses.append(np.float32(s.last_loss()).tobytes().hex())\n"
        "h = hashlib.sha256()\n"
        "for n, _, _ in s.variables():\n"
        "    h.update(s.get_param(n).tobytes())\n"
        "print('RESULT', ' '.join(losses), h.hexdigest())\n"
        "s.close()\n" % root)
    outs = []
    for graph in ("0", "1"):
        env = dict(os.environ, P3D_GRAPH=graph)
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert line, r.stdout[-2000:]
        outs.append(line[0])
        if graph == "1":
            assert "capture failed" not in r.stderr, r.stderr[-2000:]    # the capture must succeed, not quietly run eagerly
    assert outs[0] == outs[1]
