"""Parity at the reference architecture (P3D-199, 16x112x112) -- BASELINE.json configs[0..2] --
against the float32 oracle run on the GPU box's host cores."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import p3d

pytestmark = pytest.mark.gpu


def rel_l2(a, b, floor=1e-30):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), floor)


@pytest.fixture(scope="module")
def ref_params():
    return p3d.init_params(1, 'unet', None)


def test_config0_stem_and_block_a(ref_params):
    """configs[0]: single P3D-A block forward on a 1x16x112x112x3 synthetic clip, numerics only."""
    from sap3d_tensorflow_amd import P3DSession
    x = p3d.synthetic_clip(0, (1, 16, 112, 112, 3))
    _, g = p3d.forward(ref_params, x, 0.0, True, 'unet')
    s = P3DSession('unet', batch=1)
    s.load(ref_params)
    s.forward(x, 0.0, True)
    for name in ['conv1_custom', 'conv1_custom_bn_relu', 'pool1', 'block0/conv1_bn_relu', 'block0/st', 'block0/out']:
        want = g.tape.taps[name].data
        got = s.activation(name)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max(), name
        assert rel_l2(got, want) < 1e-5, name
    s.close()


def _noise_floor_check(got, o32, o64):
    """Saliency maps within 1e-3 relative of the float64 oracle -- or, where fp32 arithmetic itself
    cannot do that, no worse than 1.5x the float32 oracle's own distance to float64 (measured: the
    numpy-fp32 oracle sits 0.7..0.9e-3 from fp64 at the worst of 800k outputs of this 199-layer
    net, 2..4e-5 on average)."""
    e_hip = np.abs(got - o64) / np.abs(o64)
    e_o32 = np.abs(o32 - o64) / np.abs(o64)
    assert e_hip.max() < max(1e-3, 1.5 * e_o32.max()), (e_hip.max(), e_o32.max())
    assert e_hip.mean() < max(1e-4, 1.5 * e_o32.mean()), (e_hip.mean(), e_o32.mean())
    assert np.quantile(e_hip, 0.999) < 1e-3


@pytest.mark.parametrize("training", [False, True])
def test_config1_full_forward_b4(ref_params, training):
    """configs[1]: full backbone forward, random weights, batch 4, saliency-map difference."""
    from sap3d_tensorflow_amd import P3DSession
    x = p3d.synthetic_clip(0, (4, 16, 112, 112, 3))
    o32, _ = p3d.forward(ref_params, x, 0.0, training, 'unet')
    p64 = {k: v.astype(np.float64) for k, v in ref_params.items()}
    o64, _ = p3d.forward(p64, x.astype(np.float64), 0.0, training, 'unet', None, np.float64)
    s = P3DSession('unet', batch=4)
    s.load(ref_params)
    got = s.forward(x, 0.0, training)
    assert got.shape == (4, 16, 112, 112, 1)
    _noise_floor_check(got, o32, o64)
    s.close()


def test_config2_forward_backward(ref_params):
    """configs[2] at batch 2 (the oracle finishes in a minute): loss, saliency maps, every gradient.

    Measured on this graph (199 layers, fp32, B=2): the float32 ORACLE's gradients sit 0.14 (median) / 0.21 (max)
    rel-L2 from the float64 oracle's -- ReLU / max-pool decisions flipping on 1e-7 perturbations make deep
    gradients chaotic in fp32 -- while the loss agrees to 3e-7 and the decoder gradients to 1e-3.  So the HIP
    gradients are held to the float32 oracle's own distance from float64: median and maximum over the tensors within
    x1.5 + 2e-3, and tensor by tensor the ratio e_hip / (e_o32 + 1e-3) within twice its measured maximum
    (tests/gates.py: the two fp32 implementations flip different decisions, so single tensors differ by a factor)."""
    from sap3d_tensorflow_amd import P3DSession
    x = p3d.synthetic_clip(0, (2, 16, 112, 112, 3))
    y = p3d.synthetic_target(3, (2, 16, 112, 112))
    p64 = {k: v.astype(np.float64) for k, v in ref_params.items()}
    want_loss, want_pred, want_grads, _ = p3d.loss_and_grads(p64, x.astype(np.float64), y.astype(np.float64), 0.0, True,
                                                             'unet', None, np.float64)
    _, _, g32, _ = p3d.loss_and_grads(dict(ref_params), x, y, 0.0, True, 'unet')
    s = P3DSession('unet', batch=2)
    s.load(ref_params)
    loss, pred = s.backward(x, y, 0.0)
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss)
    e = np.abs(pred - want_pred) / np.abs(want_pred)
    assert e.max() < 1.5e-3 and e.mean() < 1e-4 and np.quantile(e, 0.999) < 1e-3, (e.max(), e.mean())
    scale = np.median([np.linalg.norm(g) for g in want_grads.values()])
    e_hip, e_o32 = {}, {}
    for n, w in want_grads.items():
        e_hip[n] = rel_l2(s.get_grad(n), w, 1e-2 * scale)
        e_o32[n] = rel_l2(g32[n], w, 1e-2 * scale)
    worst = sorted(((e_hip[n] / (e_o32[n] + 1e-3), n, e_hip[n], e_o32[n]) for n in e_hip), reverse=True)[:5]
    print("config2 gradient errors: hip median %.3e max %.3e | fp32 oracle median %.3e max %.3e | worst ratio %s" %
          (np.median(list(e_hip.values())), max(e_hip.values()), np.median(list(e_o32.values())), max(e_o32.values()), worst[0]))
    assert np.median(list(e_hip.values())) <= 1.5 * np.median(list(e_o32.values())) + 2e-3, worst
    assert max(e_hip.values()) <= 1.5 * max(e_o32.values()) + 2e-3, worst
    import gates
    gates.check("config2/worst_ratio", worst[0][0], margin=2.0, floor=0.0, detail=worst)
    for n in e_hip:      # and never beyond what a single flipped decision costs at this depth
        assert e_hip[n] <= 3.0 * e_o32[n] + 1e-2, (n, e_hip[n], e_o32[n])
    s.close()


@pytest.mark.parametrize("block_id", [0, 1, 2, 3, 4, 11, 12, 13])
def test_bottlenecks_at_batch8_against_the_oracle(ref_params, block_id):
    """Single bottlenecks of the reference architecture at the batch of configs[2] (8 clips of 16x112x112: 50176 positions in
    stage 1, 6272 in stage 2) through p3d_block_forward against the oracle's Bottleneck -- the launch shapes the train step uses:
    tiles over several rounds with K-sliced tail classes, the BatchNorm-statistics epilogue feeding the large-tensor
    BatchNorm, ST_B's sibling convs as one grouped launch, the strided first block of stage 2."""
    from oracle import nn
    from sap3d_tensorflow_amd import P3DSession
    cfg = p3d.REFERENCE_CFG
    p64 = {k: v.astype(np.float64) for k, v in ref_params.items()}
    s = P3DSession('unet', batch=8)
    s.load(ref_params)
    b = cfg.base
    stage = 0 if block_id < 3 else (1 if block_id < 11 else 2)          # blocks (3, 8, 36): stage 3 = ids 11..46 (784 positions)
    first = block_id in (0, 3, 11)
    planes = (b, 2 * b, 4 * b)[stage]
    inplanes = b if block_id == 0 else (4 * (b, 2 * b, 4 * b)[stage - 1] if first else 4 * planes)
    ishape, _ = s.block_shapes(block_id)
    assert ishape[0] == 8 and ishape[4] == inplanes
    x = np.random.default_rng(2).standard_normal(ishape).astype(np.float32)
    got = s.block_forward(block_id, x)
    s.close()
    names = list(p64)
    k0 = names.index('conv3_%d_1' % block_id)
    bn = next(n for n in names[k0:] if n.startswith('batch_normalization') and n.endswith('/gamma')).split('/')[0]
    g = p3d.Graph(p64, dtype=np.float64, create=False)
    g._uniq['batch_normalization'] = int(bn.split('_')[-1]) if '_' in bn[len('batch_normalization'):] else 0
    X = nn.Var(x.astype(np.float64))
    if first:
        want = p3d.make_block(g, X, planes, 1, inplanes, block_id, stride=2 if stage > 0 else 1).infer().data
    else:
        want = p3d.Bottleneck(g, X, inplanes, planes, n_s=block_id).infer().data
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1.0)


@pytest.mark.parametrize("block_id", [0, 1, 2, 3, 4, 11, 12, 13])
def test_bottleneck_gradients_at_batch8_against_the_oracle(ref_params, block_id):
    """The backward pass of single bottlenecks at the batch of configs[2] (p3d_block_backward): gradient of the block's input and
    of each of its variables against the float64 oracle's tape -- input gradients over several rounds with K-sliced tails and
    accumulate mode (the residual), grouped filter gradients with their cut folds, the large-tensor BatchNorm backward.
    Measured: every tensor sits where the float32 oracle itself sits (e.g. block 3: 2.9-3.7e-3 against its 3.1-4.4e-3)."""
    from oracle import nn
    from sap3d_tensorflow_amd import P3DSession
    cfg = p3d.REFERENCE_CFG
    p64 = {k: v.astype(np.float64) for k, v in ref_params.items()}
    rng = np.random.default_rng(9)
    for k, v in p64.items():                # off the symmetric initial point of the BatchNorm parameters
        if k.endswith('/gamma'):
            v[:] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('/beta'):
            v[:] = rng.uniform(-0.3, 0.3, v.shape)
    s = P3DSession('unet', batch=8)
    s.load({k: v.astype(np.float32) for k, v in p64.items()})
    b = cfg.base
    stage = 0 if block_id < 3 else (1 if block_id < 11 else 2)          # blocks (3, 8, 36): stage 3 = ids 11..46 (784 positions)
    first = block_id in (0, 3, 11)
    planes = (b, 2 * b, 4 * b)[stage]
    inplanes = b if block_id == 0 else (4 * (b, 2 * b, 4 * b)[stage - 1] if first else 4 * planes)
    ishape, oshape = s.block_shapes(block_id)
    x = rng.standard_normal(ishape).astype(np.float32)
    dy = rng.standard_normal(oshape).astype(np.float32)
    got_dx = s.block_backward(block_id, x, dy)
    names = list(p64)
    k0 = names.index('conv3_%d_1' % block_id)
    bn = next(n for n in names[k0:] if n.startswith('batch_normalization') and n.endswith('/gamma')).split('/')[0]

    def oracle(dtype):
        params = p64 if dtype == np.float64 else {k: v.astype(dtype) for k, v in p64.items()}
        g = p3d.Graph(params, dtype=dtype, create=False)
        g._uniq['batch_normalization'] = int(bn.split('_')[-1]) if '_' in bn[len('batch_normalization'):] else 0
        X = nn.Var(x.astype(dtype))
        if first:
            out = p3d.make_block(g, X, planes, 1, inplanes, block_id, stride=2 if stage > 0 else 1).infer()
        else:
            out = p3d.Bottleneck(g, X, inplanes, planes, n_s=block_id).infer()
        out.grad = dy.astype(dtype)
        for fn in reversed(g.tape.ops):
            fn()
        grads = {n: v.grad for n, v in g.trainable.items()}
        grads['(input)'] = X.grad
        return grads

    g64, g32 = oracle(np.float64), oracle(np.float32)
    assert len(g64) >= 10
    got = {n: s.get_grad(n) for n in g64 if n != '(input)'}
    got['(input)'] = got_dx
    # A ReLU / max decision that the fp32 forward takes differently from the float64 one (a few of 12.8 M pre-activations lie
    # within 1e-6 of zero) moves whole neighbourhoods of the gradient: every tensor within the fp32-noise bound of the
    # whole-graph tests -- 5 x the float32 oracle's own distance to float64 + 2e-3 -- relative to the larger of its norm and
    # 1e-2 of the median norm (conv biases in front of a batch-statistics BatchNorm have a gradient of exactly zero).
    floor = 1e-2 * np.median([np.linalg.norm(v) for v in g64.values()])
    rel = lambda a, w: np.linalg.norm(a.astype(np.float64) - w) / max(np.linalg.norm(w), floor)
    for n, w in g64.items():
        e_hip, e_o32 = rel(got[n], w), rel(g32[n], w)
        assert e_hip <= 5 * e_o32 + 2e-3, (n, e_hip, e_o32)
        print("block %d %-28s hip %.2e  fp32 oracle %.2e" % (block_id, n, e_hip, e_o32))
    s.close()


def _gn_params64(cfg):
    from oracle import p3d_gn
    p64 = p3d_gn.init_params(1, cfg, dtype=np.float64)
    rng = np.random.default_rng(7)
    for k, v in p64.items():                # off the symmetric initial point of the GroupNorm parameters
        if k.endswith('gamma'):
            v[:] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('beta'):
            v[:] = rng.uniform(-0.3, 0.3, v.shape)
    return p64


def _gn_block(params, block_id, x, dtype):
    """Bottleneck `block_id` of gn/p3d_gn.py's graph (P3D-199: blocks (3, 8, 36)) on input x, built on a fresh oracle Graph
    whose GroupNorm counter starts where the block's first group_norm scope sits in the variable list."""
    from oracle import nn, p3d_gn
    cfg = p3d.REFERENCE_CFG
    b = cfg.base
    stage = 0 if block_id < 3 else (1 if block_id < 11 else 2)
    first = block_id in (0, 3, 11)
    planes = (b, 2 * b, 4 * b)[stage]
    inplanes = b if block_id == 0 else (4 * (b, 2 * b, 4 * b)[stage - 1] if first else 4 * planes)
    names = list(params)
    k0 = names.index('conv3_%d_1' % block_id)
    gn = next(n for n in names[k0:] if n.startswith('group_norm') and n.endswith('/gamma')).split('/')[0]
    g = p3d.Graph(params, dtype=dtype, create=False)
    g._uniq['group_norm'] = int(gn.split('_')[-1]) if '_' in gn[len('group_norm'):] else 0
    X = nn.Var(x.astype(dtype))
    assert x.shape[-1] == inplanes
    if first:
        out = p3d_gn.make_block(g, X, planes, 1, inplanes, block_id, stride=2 if stage > 0 else 1).infer()
    else:
        out = p3d_gn.Bottleneck(g, X, inplanes, planes, n_s=block_id).infer()
    return g, X, out


GN_BLOCKS = [0, 1, 2, 3, 4, 11, 12]


@pytest.mark.parametrize("block_id", GN_BLOCKS)
def test_gn_cbam_bottlenecks_at_batch8_against_the_oracle(block_id):
    """configs[3]'s per-GPU share (8 clips of 16x112x112) of the GroupNorm + CBAM graph: bottlenecks of all three stages
    (gn/p3d_gn.py:127-179; 0, 3, 11 carry the projection shortcut, 3 and 11 the stride; CBAM on every residual) through
    p3d_block_forward against the oracle -- the large-tensor GroupNorm passes with their write-through hand-overs, CBAM's
    pooling / MLP / 7x7x7 kernels at 50176, 6272 and 784 positions."""
    from sap3d_tensorflow_amd import P3DSession
    p64 = _gn_params64(p3d.REFERENCE_CFG)
    s = P3DSession('gn_p3d', batch=8)
    s.load({k: v.astype(np.float32) for k, v in p64.items()})
    ishape, _ = s.block_shapes(block_id)
    x = np.random.default_rng(2).standard_normal(ishape).astype(np.float32)
    got = s.block_forward(block_id, x)
    s.close()
    want = _gn_block(p64, block_id, x, np.float64)[2].data
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1.0)


@pytest.mark.parametrize("block_id", GN_BLOCKS)
def test_gn_cbam_bottleneck_gradients_at_batch8_against_the_oracle(block_id):
    """The backward pass of the GroupNorm + CBAM bottlenecks at configs[3]'s per-GPU share (p3d_block_backward on structure
    gn_p3d; gn/p3d_gn.py:127-179, utils/network.py:198-274): the gradient of the block's input and of every one of its
    variables -- conv kernels and biases, GroupNorm gamma / beta, CBAM's shared MLP and its 7x7x7 kernel -- against the float64
    oracle's tape, each tensor within the fp32-noise bound 5 x (the float32 oracle's own distance) + 2e-3.  This is the
    full-width check of the GroupNorm backward reductions and of CBAM's backward kernels (arg-max routing of the two max
    pools, eight lanes per position in the 7x7x7 convs) at 50176, 6272 and 784 positions."""
    from sap3d_tensorflow_amd import P3DSession
    p64 = _gn_params64(p3d.REFERENCE_CFG)
    rng = np.random.default_rng(9)
    s = P3DSession('gn_p3d', batch=8)
    s.load({k: v.astype(np.float32) for k, v in p64.items()})
    ishape, oshape = s.block_shapes(block_id)
    x = rng.standard_normal(ishape).astype(np.float32)
    dy = rng.standard_normal(oshape).astype(np.float32)
    got_dx = s.block_backward(block_id, x, dy)

    def oracle(dtype):
        params = p64 if dtype == np.float64 else {k: v.astype(dtype) for k, v in p64.items()}
        g, X, out = _gn_block(params, block_id, x, dtype)
        out.grad = dy.astype(dtype)
        for fn in reversed(g.tape.ops):
            fn()
        grads = {n: v.grad for n, v in g.trainable.items()}
        grads['(input)'] = X.grad
        return grads

    g64, g32 = oracle(np.float64), oracle(np.float32)
    assert len(g64) >= 15 and any('cbam' in n for n in g64)
    got = {n: s.get_grad(n) for n in g64 if n != '(input)'}
    got['(input)'] = got_dx
    s.close()
    floor = 1e-2 * np.median([np.linalg.norm(v) for v in g64.values()])
    rel = lambda a, w: np.linalg.norm(a.astype(np.float64) - w) / max(np.linalg.norm(w), floor)
    bad = []
    for n, w in g64.items():
        e_hip, e_o32 = rel(got[n], w), rel(g32[n], w)
        print("gn block %d %-36s hip %.2e  fp32 oracle %.2e" % (block_id, n, e_hip, e_o32))
        if not e_hip <= 5 * e_o32 + 2e-3:
            bad.append((n, e_hip, e_o32))
    assert not bad, bad


@pytest.mark.parametrize("cfg,shape", [
    (p3d.NetConfig(base=16, blocks=(1, 2, 2)), (4, 16, 64, 64)),      # 5 bottlenecks: gradients are not chaotic yet, every variable is held tight
    (None, (4, 16, 112, 112)),                                      # gn/p3d_gn.py as it is (P3D-199): loss and head tight, encoder at its fp32 chaos level
])
def test_gn_shard_gradients_add_up_to_the_global_batch_gradient(cfg, shape):
    """The data-parallel contract of BASELINE configs[3] (SURVEY.md 8d cfg 4), on one GPU: GroupNorm and CBAM normalise and
    pool per clip, the loss is a batch SUM (utils/network.py:60), so the gradient of the global batch is the sum of the
    per-shard gradients -- which is all the all-reduce adds up.  HIP gn_p3d: backward of clips [0,2) plus backward of clips
    [2,4) (what two ranks would hold) against one backward of the four clips; the two losses add up to 2e-6.
    No clip's forward depends on its batch mates (no batch statistics anywhere in this graph), so only the ORDER of the sums
    differs between the runs (other tile / K-slice plans at 2 and 4 clips).  At 5 bottlenecks that is fp32 rounding for every
    variable.  Through the 47 bottlenecks of the reference architecture a last-bit difference flips ReLU / arg-max decisions
    and the encoder gradients of two fp32 evaluations differ by per cent (measured here: median 2.7e-2, worst 9.8e-2 -- the
    float32 ORACLE sits 0.14 / 0.21 from the float64 one on this depth, test_config2_forward_backward); there the loss and the
    head's variables (behind the encoder's stable forward) carry the check."""
    from sap3d_tensorflow_amd import P3DSession
    from sap3d_tensorflow_amd import synthetic
    x = synthetic.synthetic_clip(0, shape + (3,))
    y = synthetic.synthetic_target(3, shape)
    kw = dict(frames=shape[1], height=shape[2], width=shape[3])
    if cfg is not None:
        kw.update(base=cfg.base, blocks=cfg.blocks)
    whole = P3DSession('gn_p3d', batch=4, seed=5, **kw)
    theta = whole.save()
    l4, _ = whole.backward(x, y, 0.0)
    names = [n for n, _, t in whole.variables() if t]
    g4 = {n: whole.get_grad(n).astype(np.float64) for n in names}
    whole.close()
    half = P3DSession('gn_p3d', batch=2, seed=99, **kw)
    half.load(theta)
    gs = {n: 0.0 for n in names}
    ls = 0.0
    for lo in (0, 2):
        l2, _ = half.backward(x[lo:lo + 2], y[lo:lo + 2], 0.0)
        ls += float(l2)
        for n in names:
            gs[n] = gs[n] + half.get_grad(n).astype(np.float64)
    half.close()
    assert abs(ls - l4) <= 2e-6 * abs(l4), (ls, l4)
    scale = np.median([np.linalg.norm(v) for v in g4.values()])
    errs = {n: np.linalg.norm(gs[n] - g4[n]) / max(np.linalg.norm(g4[n]), 1e-2 * scale) for n in names}
    worst = sorted(((e, n) for n, e in errs.items()), reverse=True)[:5]
    print("shard sum vs global batch %s: median %.2e worst %s" % (shape, np.median(list(errs.values())), worst))
    head = [n for n in names if any(k in n for k in ('deconv_pool4', 'conv_concat', 'deconv_revise', 'predict_revise'))]
    assert len(head) >= 8, head
    if cfg is not None:
        assert worst[0][0] <= 2e-4, worst
    else:
        assert max(errs[n] for n in head) <= 2e-3, sorted(((errs[n], n) for n in head), reverse=True)[:3]
        assert np.median(list(errs.values())) <= 0.1 and worst[0][0] <= 0.3, worst


@pytest.mark.parametrize("structure,shape,step,tol,first_group", [
    ("unet", (8, 16, 112, 112), 2e-5, 8e-2, 0),       # BASELINE.json configs[2]: batch 8, 16x112x112
    ("unet", (1, 32, 224, 224), 2e-5, 8e-2, 0),       # the clip shape of configs[4] (32 frames of 224x224), one clip
    # configs[3] graph (GroupNorm + CBAM): CBAM's arg-max routing bends the loss sooner, so half the step.  All four groups
    # (the kernels are bit-reproducible since round 2, so the stem + stage-1 group is measurable too), at two clips and at
    # configs[3]'s per-GPU share of eight.
    ("gn_p3d", (2, 16, 112, 112), 1e-5, 15e-2, 0),
    ("gn_p3d", (8, 16, 112, 112), 1e-5, 15e-2, 0),
])
def test_directional_derivative_at_full_size(structure, shape, step, tol, first_group):
    """Size-independent property, no oracle involved: moving the parameters by a small step delta must change the
    loss by <g, delta>, g being the gradient the backward pass returned.  Checked at the reference architecture
    (62 M parameters, 199 layers) and at BASELINE clip sizes where the numpy oracle would need hours, separately
    for four parameter groups (stem + stage 1, stage 2, stage 3, head) with delta along the group's own gradient.
    The step is tiny on purpose: at random initialisation the loss of this 199-layer ReLU/max-pool net is linear
    along a backbone direction only while it moves by ~1e-4 of its value (tools/archive/dd_probe.py: the predicted change
    is met within 1.5 % at 2e-5 L and saturates beyond 3e-4 L; head directions stay linear 1000x further).  So
    every group moves the loss by `step` = 1-2e-5 L (fp32 read-back of L resolves ~1e-7 L), central difference, and the
    prediction uses the step each fp32 parameter actually took, most components being below one ulp."""
    from sap3d_tensorflow_amd import P3DSession
    from sap3d_tensorflow_amd import synthetic
    import re
    B, T, H, W = shape
    s = P3DSession(structure, batch=B, frames=T, height=H, width=W, seed=3)
    x = synthetic.synthetic_clip(0, (B, T, H, W, 3))
    y = synthetic.synthetic_target(3, (B, T, H, W))
    theta = s.save()
    rng = np.random.default_rng(0)
    for n in sorted(theta):                # move off the symmetric point beta = 0 (about half of every layer active)
        if n.endswith('/beta'):
            theta[n] = rng.uniform(-0.2, 0.2, theta[n].shape).astype(np.float32)
    s.load(theta)
    loss0, _ = s.backward(x, y, 0.0)
    assert np.isfinite(loss0)
    trainable = [n for n, _, t in s.variables() if t]
    g = {n: s.get_grad(n).astype(np.float64) for n in trainable}
    # variables are created in forward order: cut the list at the first variable of blocks 3 and 11 and of the head
    order = {n: i for i, n in enumerate(trainable)}
    def first(pattern):
        return min(i for n, i in order.items() if re.search(pattern, n))
    cuts = [0, first(r'conv3_3_1$'), first(r'conv3_11_1$'),
            first(r'^(conv3d_transpose/kernel|deconv_pool3/kernel)$'), len(trainable)]
    if structure == 'gn_p3d':              # deconv_pool3 is created before stage 3 there: head = everything after stage 3
        cuts[3] = first(r'^deconv_pool4/kernel$')
    groups = [trainable[cuts[k]:cuts[k + 1]] for k in range(4)]
    assert all(len(grp) > 10 for grp in groups) and sum(len(grp) for grp in groups) == len(trainable)
    for k, names in enumerate(groups):
        if k < first_group:
            continue
        gn = float(np.sqrt(sum((g[n] ** 2).sum() for n in names)))
        assert np.isfinite(gn) and gn > 0, k
        # The loss is piecewise smooth (ReLU, max-pool, CBAM's arg-max): one decision that flips between theta + delta and
        # theta - delta moves a central difference by tens of per cent, and WHICH step lands on one depends on the last bit of
        # the forward pass (it changed sides when the K-slice plan of a decoder conv changed).  Three step sizes, the median
        # deviation decides.  (Round 4: this property is no longer what carries the gradient claim at these sizes -- the
        # GroupNorm + CBAM bottlenecks are compared with the oracle's gradients directly at batch 8, 1e-7-level agreement on
        # every tensor (test_gn_cbam_bottleneck_gradients_at_batch8_against_the_oracle), and tests/test_gpu_pinned.py shows what
        # a flipped decision does to a whole-graph comparison: 1e-2 unpinned, 1e-4 on the same branch.)
        devs = []
        for scale in (1.0, 1.5, 2.0):
            eps = scale * step * abs(loss0) / gn
            losses, moved = [], []
            for sign in (+1.0, -1.0):
                m = dict(theta)
                for n in names:
                    m[n] = (theta[n].astype(np.float64) + sign * eps * g[n] / gn).astype(np.float32)
                s.load(m)
                losses.append(s.backward(x, y, 0.0)[0])
                moved.append(m)
            predicted = sum(float((g[n] * (moved[0][n].astype(np.float64) - moved[1][n].astype(np.float64))).sum()) for n in names)
            measured = losses[0] - losses[1]
            assert predicted > 0.5 * scale * step * abs(loss0), (k, predicted)
            devs.append((abs(measured - predicted) / abs(predicted), scale, measured, predicted))
        devs.sort()
        assert devs[1][0] <= tol, (k, devs, loss0, gn)
    s.close()


def test_pointwise_fp16_at_reference_size(ref_params):
    """BASELINE.json configs[4] arithmetic (fp16 MFMA pointwise convs, fp32 accumulate) on the reference architecture.
    At random initialisation this 199-layer net amplifies any perturbation ~1.2x per bottleneck (tools/f16_probe.py:
    the 5e-4 fp16 rounding is 2 % rel-L2 at block 10, 55 % at block 46, and the skip connections of the decoder
    bring the saliency maps back to 4 % rel-L2; the same growth turns fp32's 1e-7 into the 1e-3 noise floor of the
    fp32 tests).  So SURVEY.md 8d's "2e-2 relative on pred" can hold for the bulk of the map, not for its maximum:
    mean relative deviation from the fp32 path <= 2e-2, 99th percentile <= 1e-1, loss within 1e-3."""
    from sap3d_tensorflow_amd import P3DSession
    x = p3d.synthetic_clip(0, (2, 16, 112, 112, 3))
    y = p3d.synthetic_target(3, (2, 16, 112, 112))
    s = P3DSession('unet', batch=2)
    s.load(ref_params)
    full = s.forward(x, 0.0, True)
    l32, _ = s.backward(x, y, 0.0)
    s.set_pointwise_fp16(True)
    half = s.forward(x, 0.0, True)
    l16, _ = s.backward(x, y, 0.0)
    s.close()
    rel = np.abs(half - full) / np.abs(full)
    assert np.isfinite(half).all() and np.isfinite(l16)
    assert rel.mean() <= 2e-2, rel.mean()
    assert np.quantile(rel, 0.99) <= 1e-1
    assert rel.max() > 1e-6
    assert abs(l16 - l32) <= 1e-3 * abs(l32)


def test_configs4_workload_fp16_pointwise_batch8():
    """BASELINE.json configs[4] as ONE GPU's share of it: 8 clips of 32x224x224, pointwise convs on the fp16 MFMA.
    (The 8-GPU weak-scaling curve is the driver's to measure.)  Checked at that size: (i) the fp16-mode saliency maps and
    loss against the fp32 HIP path on the same weights -- fp16-level, as in test_pointwise_fp16_at_reference_size;
    (ii) two train steps run and lower nothing to NaN; (iii) the bit-reproducibility of the step holds in this mode too.
    Operands stay fp32 in HBM in this mode (rounded to fp16 in registers): it is configs[4]'s arithmetic, not yet its
    storage format."""
    from sap3d_tensorflow_amd import P3DSession
    from sap3d_tensorflow_amd import synthetic
    shape = (8, 32, 224, 224)
    s = P3DSession('unet', batch=shape[0], frames=shape[1], height=shape[2], width=shape[3], seed=1)
    x = synthetic.synthetic_clip(0, shape + (3,))
    y = synthetic.synthetic_target(3, shape)
    l32, full = s.backward(x, y, 0.0)
    s.set_pointwise_fp16(True)
    l16, half = s.backward(x, y, 0.0)
    g_a = s.get_grad('conv3_20_1')
    l16b, _ = s.backward(x, y, 0.0)
    assert np.float32(l16).tobytes() == np.float32(l16b).tobytes() and np.array_equal(g_a, s.get_grad('conv3_20_1'))
    assert half.shape == shape + (1,) and np.isfinite(half).all() and np.isfinite(l16)
    rel = np.abs(half - full) / np.abs(full)
    assert rel.mean() <= 2e-2, rel.mean()
    assert np.quantile(rel, 0.99) <= 1e-1
    assert rel.max() > 1e-6                      # the mode really changed the arithmetic
    assert abs(l16 - l32) <= 1e-3 * abs(l32)
    a = s.train_step(x, y, dropout=0.5, seed=1)
    b = s.train_step(x, y, dropout=0.5, seed=2)
    assert np.isfinite(a) and np.isfinite(b)
    s.close()


def test_gn_head_at_full_width_against_the_oracle():
    """The head of inference_p3d (gn/p3d_gn.py:234-257) at the REFERENCE widths, inside the train step's own schedule: base 64
    with one bottleneck per stage keeps the float64 oracle affordable while deconv_pool3 (512 -> 512), deconv_pool4 (kernel 3,
    stride 4, 1024 -> 1024), conv_concat (3x3x3, 1792 -> 1024 on 4x28x28: 311 of the net's 382 GFLOP per clip), deconv_revise
    (1024 -> 256) and predict_revise (256 -> 1) run at full size on 2 clips.  Map and loss against float64; every HEAD gradient
    (kernels, biases, GroupNorm gamma / beta) within the plain fp32-noise bound 5 x (float32 oracle's distance) + 2e-3, no
    measured allowance; the three encoder bottlenecks (CBAM arg-max routing) go through the measured gate."""
    from oracle import p3d_gn
    from gates import check, noise_excess
    from sap3d_tensorflow_amd import P3DSession
    cfg = p3d.NetConfig(base=64, blocks=(1, 1, 1))
    shape = (2, 16, 112, 112)
    p64 = _gn_params64(cfg)
    p32 = {k: v.astype(np.float32) for k, v in p64.items()}
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    s = P3DSession('gn_p3d', batch=shape[0], frames=shape[1], height=shape[2], width=shape[3], base=cfg.base, blocks=cfg.blocks)
    s.load(p32)
    assert [n for n, _, _ in s.variables()] == list(p64)
    l64, pr64, g64, graph = p3d_gn.loss_and_grads(p64, x.astype(np.float64), y.astype(np.float64), 0.0, True, cfg, np.float64)
    loss, pred = s.backward(x, y, 0.0)
    for name in ['conv_concat']:
        want = graph.tape.taps[name].data
        got = s.activation(name)
        assert got.shape == want.shape, name
        assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1.0), name
    assert np.abs(pred - pr64).max() <= 1e-4 * max(np.abs(pr64).max(), 1.0)
    assert abs(loss - l64) < 1e-5 * abs(l64)
    _, _, g32, _ = p3d_gn.loss_and_grads(dict(p32), x, y, 0.0, True, cfg, np.float32)
    floor = 1e-2 * np.median([np.linalg.norm(v) for v in g64.values()])
    e_hip = {n: rel_l2(s.get_grad(n), w, floor) for n, w in g64.items()}
    e_o32 = {n: rel_l2(g32[n], w, floor) for n, w in g64.items()}
    s.close()
    names = list(g64)
    head0 = names.index('deconv_pool4/kernel')
    head = [n for n in names[head0:]] + [n for n in names if n.startswith('deconv_pool3')]
    # GroupNorm scopes are numbered in creation order: the ones created after deconv_pool4 belong to the head, and
    # deconv_pool3's is the one right behind its bias
    k3 = names.index('deconv_pool3/bias')
    head += [n for n in names[k3 + 1:k3 + 3] if n.startswith('group_norm')]
    assert any(n == 'conv_concat/kernel' for n in head) and any(n == 'predict_revise/kernel' for n in head) and len(head) >= 18
    worst, who = noise_excess({n: e_hip[n] for n in head}, e_o32)
    assert worst == 0.0, ("a head gradient lies above the fp32-noise bound", who, e_hip[who], e_o32[who])
    rest = {n: e for n, e in e_hip.items() if n not in head}
    excess, name = noise_excess(rest, e_o32)
    check("gn_head_full/encoder", excess, detail=(name, rest.get(name), e_o32.get(name)))
