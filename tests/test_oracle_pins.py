"""The decision pins of the oracle (oracle/nn.py Tape.pins), on CPU: pinned to its OWN decisions an evaluation reproduces itself
bit for bit; pinned to another evaluation's decisions (float32's, here) the float64 gradient is the gradient of that other
branch -- it moves away from the unpinned float64 gradient exactly when decisions differ, and two evaluations pinned to the
same decisions agree to rounding, whatever their own arithmetic would have decided."""
import numpy as np

from oracle import nn, p3d


def _own_decisions(graph):
    """{'relu': {scope: mask}, 'pool': [inputs]} read off a finished evaluation's tape (what P3DSession.decisions() returns
    for a HIP pass)."""
    return graph.tape.own


def _run(params, x, y, cfg, dtype, pins=None, record=False):
    g = p3d.Graph({k: v.astype(dtype) for k, v in params.items()}, dtype=dtype, create=False)
    g.tape.pins = pins
    if record:
        g.tape.own = {"relu": {}, "pool": []}
        relu0, pool0 = nn.relu, nn.max_pool3d

        def relu(tape, v):
            out = relu0(tape, v)
            if v.tag is not None:
                tape.own["relu"][v.tag] = v.data > 0
            return out

        def pool(tape, v, k, s):
            tape.own["pool"].append(v.data.copy())
            return pool0(tape, v, k, s)
        nn.relu, nn.max_pool3d = relu, pool
    try:
        X = nn.Var(x.astype(dtype))
        pred = p3d.STRUCTURES['unet'](g, X, 0.0, x.shape[0], True, cfg, None)
        loss = nn.smooth_l1_loss(g.tape, nn.reshape(g.tape, pred, y.shape), y.astype(dtype), 1, 1, sigma=1.0)
        g.tape.backward(loss)
    finally:
        if record:
            nn.relu, nn.max_pool3d = relu0, pool0
    return float(loss.data), {n: v.grad for n, v in g.trainable.items()}, g


def test_pins_reproduce_and_transfer_decisions():
    cfg = p3d.NetConfig(base=8, blocks=(1, 1, 2))
    shape = (1, 16, 32, 32)
    params = p3d.init_params(1, 'unet', cfg, dtype=np.float64)
    rng = np.random.default_rng(3)
    for k, v in params.items():
        if k.endswith('/beta'):
            v[:] = rng.uniform(-0.3, 0.3, v.shape)
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    l64, g64, gr64 = _run(params, x, y, cfg, np.float64, record=True)
    own = _own_decisions(gr64)
    assert len(own["relu"]) >= 15 and len(own["pool"]) == 4
    # pinned to its own decisions: the same evaluation, bit for bit, and nothing counted as flipped
    l_same, g_same, gr_same = _run(params, x, y, cfg, np.float64, pins=own)
    assert l_same == l64
    assert all(np.array_equal(g_same[n], g64[n]) for n in g64)
    assert gr_same.tape.pin_log["relu"] == len(own["relu"]) and gr_same.tape.pin_log["relu_flips"] == 0
    assert gr_same.tape.pin_log["pool"] == 4 and gr_same.tape.pin_log["pool_flips"] == 0
    # a flipped gate changes the gradient (the pins are really used) ...
    bent = {"relu": dict(own["relu"]), "pool": []}        # (no pool pins: downstream of 64 forced gates the pools' inputs are other tensors)
    name = sorted(bent["relu"])[3]
    m = bent["relu"][name].copy()
    m.reshape(-1)[:64] ^= True
    bent["relu"][name] = m
    _, g_bent, gr_bent = _run(params, x, y, cfg, np.float64, pins=bent)
    assert gr_bent.tape.pin_log["relu_flips"] >= 64          # the forced ones, and downstream gates that the changed values would now take otherwise
    assert max(np.abs(g_bent[n] - g64[n]).max() for n in g64) > 0
    # ... and a float32 evaluation pinned to the float64 decisions agrees with it to float32 rounding on EVERY tensor (unpinned,
    # a flipped near-zero decision would show as per cent on some)
    _, g32p, _ = _run(params, x, y, cfg, np.float32, pins=own)
    scale = np.median([np.linalg.norm(v) for v in g64.values()])
    worst = max(np.linalg.norm(g32p[n].astype(np.float64) - g64[n]) / max(np.linalg.norm(g64[n]), 1e-2 * scale) for n in g64)
    assert worst < 2e-4, worst
