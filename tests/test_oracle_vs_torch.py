"""The oracle (oracle/, numpy restatement of the TF-1.x graph) against an
independent torch-CPU composition of the same graph with autograd.  The
reference itself cannot run here and holds no fixtures (parity
unpinned); this is what pins the oracle instead."""
import numpy as np
import pytest
import torch

from oracle import nn, p3d
import torch_ref


def _randomise_norm_params(params, seed=5):
    rng = np.random.default_rng(seed)
    for k, v in params.items():
        if k.startswith('gamma'):           # attention mixing scalars (init 0 would switch their branch's gradients off)
            v[:] = rng.uniform(0.4, 1.0, v.shape)
        elif k.endswith('gamma'):
            v[:] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('beta'):
            v[:] = rng.uniform(-0.3, 0.3, v.shape)
        elif k.endswith('moving_mean'):
            v[:] = rng.uniform(-0.1, 0.1, v.shape)
        elif k.endswith('moving_variance'):
            v[:] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('/bias'):
            v[:] = rng.uniform(-0.1, 0.1, v.shape)
    return params


def rel_l2(a, b, floor):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), floor)


TORCH_METHOD = {"unet": "unet", "concat": "concat", "unet++nonsa": "unetpp_nonsa", "unet++ds": "unetpp_ds"}


@pytest.mark.parametrize("structure", ["unet", "concat", "unet++nonsa", "unet++ds"])
@pytest.mark.parametrize("training", [True, False])
def test_small_net_fp64(structure, training):
    cfg = p3d.NetConfig(base=8, blocks=(3, 3, 3))
    params = _randomise_norm_params(p3d.init_params(1, structure, cfg, dtype=np.float64))
    x = p3d.synthetic_clip(0, (2, 16, 32, 32, 3)).astype(np.float64)
    y = p3d.synthetic_target(3, (2, 16, 32, 32)).astype(np.float64)
    loss, pred, grads, g = p3d.loss_and_grads(params, x, y, 0.0, training, structure, cfg, np.float64)
    m = torch_ref.TorchP3D(params, torch.float64, cfg.base, cfg.blocks)
    tp = getattr(m, TORCH_METHOD[structure])(torch.tensor(x), training)
    tl = torch_ref.smooth_l1_sum(tp.reshape(y.shape), torch.tensor(y))
    tl.backward()
    assert abs(loss - tl.item()) <= 1e-10 * abs(tl.item())
    assert np.abs(pred - tp.detach().numpy()).max() < 1e-11
    # a bias in front of a batch-statistics BN has an exactly-zero gradient: use
    # an absolute floor tied to the typical gradient size for those
    scale = np.median([np.linalg.norm(gr) for gr in grads.values()])
    for n, gr in grads.items():
        tg = m.p[n].grad.numpy()
        assert rel_l2(gr, tg, 1e-4 * scale) < 1e-8, n
    # BN moving statistics (UPDATE_OPS)
    g.tape.apply_updates()
    for n, v in m.new_moving.items():
        assert np.allclose(params[n], v.numpy(), rtol=1e-12, atol=1e-14), n


def test_small_net_fp32_matches_fp64():
    """fp32 oracle vs fp64 oracle = the noise floor any fp32 implementation has.
    Forward values are tight (1e-5); gradients are NOT: a ReLU / max-pool decision
    that flips on a 1e-7 perturbation changes one gradient element by O(1), so
    two correct fp32 implementations differ by ~sqrt(flip fraction) ~ 1e-3..1e-2
    in rel-L2 on deep gradients.  GPU parity tests therefore judge gradients
    against this floor, not against 1e-3."""
    cfg = p3d.NetConfig(base=8, blocks=(3, 3, 3))
    p64 = _randomise_norm_params(p3d.init_params(1, 'unet', cfg, dtype=np.float64))
    p32 = {k: v.astype(np.float32) for k, v in p64.items()}
    x = p3d.synthetic_clip(0, (2, 16, 32, 32, 3))
    y = p3d.synthetic_target(3, (2, 16, 32, 32))
    l64, pr64, g64, _ = p3d.loss_and_grads(p64, x.astype(np.float64), y.astype(np.float64), 0.0, True, 'unet', cfg, np.float64)
    l32, pr32, g32, _ = p3d.loss_and_grads(p32, x, y, 0.0, True, 'unet', cfg, np.float32)
    assert abs(l32 - l64) < 1e-6 * abs(l64)
    assert np.abs(pr32 - pr64).max() < 1e-4
    scale = np.median([np.linalg.norm(gr) for gr in g64.values()])
    errs = [rel_l2(g32[n].astype(np.float64), g64[n], 1e-2 * scale) for n in g64]
    assert np.median(errs) < 2e-2 and max(errs) < 0.2


@pytest.mark.parametrize("k,s", [(3, 2), (1, 2), (2, 2), (3, 4), (3, 1)])
def test_deconv_same_closed_form(k, s):
    """Appendix A.3 closed form per axis: full[o] = sum x[i] W[kk] at o=i*s+kk,
    drop max(k-s,0)//2 from the start, crop / zero-extend to in*s."""
    rng = np.random.default_rng(0)
    I = 5
    x = rng.standard_normal((1, I, 1, 1, 1))
    w = rng.standard_normal((k, 1, 1, 1, 1))
    t = nn.Tape()
    y = nn.conv3d_transpose(t, nn.Var(x), nn.Var(w), (s, 1, 1)).data[0, :, 0, 0, 0]
    full = np.zeros((I - 1) * s + k)
    for i in range(I):
        for kk in range(k):
            full[i * s + kk] += x[0, i, 0, 0, 0] * w[kk, 0, 0, 0, 0]
    start = max(k - s, 0) // 2
    want = np.zeros(I * s)
    seg = full[start:start + I * s]
    want[:len(seg)] = seg
    assert np.allclose(y, want, atol=1e-14)


def test_same_padding_instances():
    """Appendix A.1 worked instances."""
    assert nn.same_pads(112, 7, 2) == (56, 2, 3)
    assert nn.same_pads(56, 3, 2) == (28, 0, 1)
    assert nn.same_pads(16, 2, 2) == (8, 0, 0)
    assert nn.same_pads(28, 1, 2) == (14, 0, 0)
    assert nn.same_pads(28, 3, 1) == (28, 1, 1)


def test_adam_matches_tf_formula():
    rng = np.random.default_rng(0)
    p = rng.standard_normal(7)
    g = rng.standard_normal(7)
    m = np.zeros(7)
    v = np.zeros(7)
    p0 = p.copy()
    nn.adam_step(p, g, m, v, 1, lr=1e-4)
    # first step: m=0.1g, v=0.001g^2, lr_t = lr*sqrt(.001)/.1
    want = p0 - 1e-4 * np.sqrt(0.001) / 0.1 * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    assert np.allclose(p, want, rtol=1e-12)


def test_param_inventory_reference_arch():
    """61 943 105 trainables, 197 conv/deconv kernels, 195 norm layers
    (SURVEY.md Appendix B totals)."""
    params = p3d.init_params(1, 'unet', None)
    train = {k: v for k, v in params.items() if not k.endswith(('moving_mean', 'moving_variance'))}
    assert sum(v.size for v in train.values()) == 61943105
    assert sum(1 for k in params if k.endswith('moving_mean')) == 195
    assert sum(1 for k, v in train.items() if v.ndim == 5) == 197
    assert params['batch_normalization_191/gamma'].shape == (1024,)


def test_gn_cbam_net_fp64():
    """gn/p3d_gn.py inference_p3d (GroupNorm, CBAM on every residual, concat head) vs torch autograd."""
    from oracle import p3d_gn
    cfg = p3d.NetConfig(base=8, blocks=(2, 2, 2))
    params = p3d_gn.init_params(1, cfg, dtype=np.float64)
    rng = np.random.default_rng(7)
    for k, v in params.items():
        if k.endswith('gamma'):
            v[:] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith(('beta', '/bias')):
            v[:] = rng.uniform(-0.3, 0.3, v.shape)
    x = p3d.synthetic_clip(0, (2, 16, 32, 32, 3)).astype(np.float64)
    y = p3d.synthetic_target(3, (2, 16, 32, 32)).astype(np.float64)
    loss, pred, grads, g = p3d_gn.loss_and_grads(params, x, y, 0.0, True, cfg, np.float64)
    m = torch_ref.TorchP3DGN(params, torch.float64, cfg.base, cfg.blocks)
    tp = m.inference_p3d(torch.tensor(x))
    tl = torch_ref.smooth_l1_sum(tp.reshape(y.shape), torch.tensor(y))
    tl.backward()
    assert abs(loss - tl.item()) <= 1e-10 * abs(tl.item())
    assert np.abs(pred - tp.detach().numpy()).max() < 1e-10
    scale = np.median([np.linalg.norm(gr) for gr in grads.values()])
    assert len(grads) == len([k for k in m.p if m.p[k].requires_grad])
    for n, gr in grads.items():
        assert rel_l2(gr, m.p[n].grad.numpy(), 1e-4 * scale) < 1e-8, n


def test_gn_decoder_block_net_fp64():
    """gn/p3d_gn.py:489 inference_p3d_decoder_block (net='P3D_DECODER': skip deconvs with k<s, two decoder
    blocks, stride-1 conv to one channel, variables under scope 'P3D/') vs torch autograd."""
    from oracle import p3d_gn
    cfg = p3d.NetConfig(base=16, blocks=(1, 2, 2))
    params = p3d_gn.init_params(1, cfg, dtype=np.float64, head='decoder')
    assert all(k.startswith('P3D/') for k in params)
    assert 'P3D/results/kernel' in params and params['P3D/results/kernel'].shape == (3, 3, 3, 4, 1)
    assert params['P3D/deconv_pool4/kernel'].shape == (1, 3, 3, 128, 256)
    rng = np.random.default_rng(7)
    for k, v in params.items():
        if k.endswith('gamma'):
            v[:] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith(('beta', '/bias')):
            v[:] = rng.uniform(-0.3, 0.3, v.shape)
    x = p3d.synthetic_clip(0, (1, 16, 32, 32, 3)).astype(np.float64)
    y = p3d.synthetic_target(3, (1, 16, 32, 32)).astype(np.float64)
    loss, pred, grads, g = p3d_gn.loss_and_grads(params, x, y, 0.0, True, cfg, np.float64, head='decoder')
    bare = {k[len('P3D/'):]: v for k, v in params.items()}
    m = torch_ref.TorchP3DGN(bare, torch.float64, cfg.base, cfg.blocks)
    tp = m.decoder_block(torch.tensor(x))
    tl = torch_ref.smooth_l1_sum(tp.reshape(y.shape), torch.tensor(y))
    tl.backward()
    assert abs(loss - tl.item()) <= 1e-10 * abs(tl.item())
    assert np.abs(pred - tp.detach().numpy()).max() < 1e-10
    scale = np.median([np.linalg.norm(gr) for gr in grads.values()])
    assert len(grads) == len([k for k in m.p if m.p[k].requires_grad])
    for n, gr in grads.items():
        assert rel_l2(gr, m.p[n[len('P3D/'):]].grad.numpy(), 1e-4 * scale) < 1e-8, n
