"""BatchNorm fusion (conv_igemm2.hip / conv_wgrad2.hip operand transforms) against the unfused launch list of the same
library AND against the oracle: the two paths compute tf.layers.batch_normalization's arithmetic (p3d.py:56-81,88-97) with
different summation orders only, so they must agree far inside the oracle gates."""
import numpy as np
import pytest

from oracle import p3d

pytestmark = pytest.mark.gpu


def rel_l2(a, b, floor):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), floor)


def session(cfg, shape, params, structure="unet"):
    from sap3d_tensorflow_amd import P3DSession
    B, T, H, W = shape
    s = P3DSession(structure, batch=B, frames=T, height=H, width=W, base=cfg.base, blocks=cfg.blocks)
    s.load(params)
    return s


def randomise(params, seed=5):
    rng = np.random.default_rng(seed)
    for k, v in params.items():
        if k.endswith('gamma'):
            v[:] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('beta'):
            v[:] = rng.uniform(-0.3, 0.3, v.shape)
    return params


# (config, clip shape): few statistics partials (folded in the consumers' prologues), many (finalize launches: stage 1 of the
# third case has 18432 rows = 144-288 partials), all three bottleneck types in every stage, strided first blocks
CASES = [
    (p3d.NetConfig(base=8, blocks=(3, 3, 3)), (2, 16, 32, 32)),
    (p3d.NetConfig(base=16, blocks=(1, 2, 4)), (1, 16, 48, 32)),
    (p3d.NetConfig(base=16, blocks=(3, 3, 3)), (4, 16, 96, 96)),
]


@pytest.mark.parametrize("cfg,shape", CASES)
def test_fused_equals_unfused(cfg, shape):
    params = randomise(p3d.init_params(1, 'unet', cfg))
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    s = session(cfg, shape, params)
    names = [n for n, _, tr in s.variables() if tr]
    out = {}
    for mode in (0, 1, 2):       # passes of their own / forward fused / forward and backward fused
        s.set_bn_fusion(mode)
        loss, pred = s.backward(x, y, 0.0)
        taps = {n: s.activation(n) for n in ('block0/conv1_bn_relu', 'block0/st', 'block1/st', 'block2/st', 'block2/out', 'pool4')}
        out[mode] = (loss, pred, {n: s.get_grad(n) for n in names}, taps)
    l0, p0, g0, t0 = out[0]
    scale = np.median([np.linalg.norm(g) for g in g0.values()])
    for mode in (1, 2):
        l1, p1, g1, t1 = out[mode]
        assert abs(l1 - l0) <= 1e-5 * abs(l0)
        assert np.abs(p1 - p0).max() <= 2e-5
        for n in t0:
            assert np.abs(t1[n] - t0[n]).max() <= 1e-4 * max(1.0, np.abs(t0[n]).max()), (mode, n)
        # the two launch lists differ in summation order only; where that flips ONE ReLU decision of these small nets, a
        # 16-32 element BatchNorm gradient moves by a few 1e-3 (measured up to 4e-3); a wrong transform shows as >= 1e-1
        worst = max((rel_l2(g1[n], g0[n], 1e-2 * scale), n) for n in names)
        assert worst[0] <= 1e-2, (mode, worst)
        errs = sorted(rel_l2(g1[n], g0[n], 1e-2 * scale) for n in names)
        assert errs[len(errs) // 2] <= 5e-4, (mode, errs[len(errs) // 2])
    s.close()


def test_fused_train_steps_match_unfused_and_oracle():
    cfg, shape = CASES[0]
    p64 = randomise(p3d.init_params(1, 'unet', cfg, dtype=np.float64))
    p32 = {k: v.astype(np.float32) for k, v in p64.items()}
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    runs = {}
    for mode in (0, 2):
        s = session(cfg, shape, p32)
        s.set_bn_fusion(mode)
        s.set_adam(1e-3)
        losses = [s.train_step(x, y, dropout=0.0) for _ in range(3)]
        runs[mode] = (losses, {n: s.get_param(n) for n, _, _ in s.variables() if n.endswith(('moving_mean', 'moving_variance'))})
        s.close()
    state = {'t': 0, 'm': {}, 'v': {}}
    want = [p3d.train_step(p64, state, x.astype(np.float64), y.astype(np.float64), lr=1e-3, cfg=cfg, dtype=np.float64)[0] for _ in range(3)]
    for it in range(3):
        # Adam's first steps are lr * sign(g): weights whose gradient is noise-level take a different +-lr step than the
        # oracle's, which the third loss shows at the 2e-4 level for BOTH launch lists (tests/archive/probes/fused_probe.py)
        assert abs(runs[2][0][it] - want[it]) < 5e-4 * abs(want[it]), (it, runs[2][0][it], want[it])
        assert abs(runs[2][0][it] - runs[0][0][it]) < 5e-4 * abs(want[it])      # (the same +-lr steps: the two launch lists round differently)
    for n, v in runs[2][1].items():
        assert np.allclose(v, p64[n], rtol=1e-2, atol=2e-3), n
        assert np.allclose(v, runs[0][1][n], rtol=1e-3, atol=1e-4), n


def test_fused_backward_is_bit_reproducible():
    cfg, shape = CASES[1]
    params = randomise(p3d.init_params(1, 'unet', cfg))
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    s = session(cfg, shape, params)
    s.set_bn_fusion(2)
    names = [n for n, _, tr in s.variables() if tr]
    l0, p0 = s.backward(x, y, 0.0)
    g0 = {n: s.get_grad(n) for n in names}
    l1, p1 = s.backward(x, y, 0.0)
    assert l0 == l1 and np.array_equal(p0, p1)
    for n in names:
        assert np.array_equal(g0[n], s.get_grad(n)), n
    s.close()
