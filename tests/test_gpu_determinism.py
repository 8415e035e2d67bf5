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
