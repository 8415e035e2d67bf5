"""Generates tests/golden/*.npz from the oracle (oracle/, float64 run, stored as float32).

The reference cannot run here, so these vectors pin the ORACLE's behaviour (so that
a later edit to oracle/ cannot silently move the target) and let the GPU parity tests run against
committed numbers.  Inputs are regenerated from seeds; only expected outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import p3d, p3d_gn      # noqa: E402

CASES = {
    # name: (base, blocks, (B,T,H,W))            -- p3d_unet, gradients listed in GRADS
    "unet_b8_333": (8, (3, 3, 3), (2, 16, 32, 32)),
    "unet_b16_124": (16, (1, 2, 4), (1, 16, 48, 32)),
}
# the other graphs: name -> (structure, base, blocks, (B,T,H,W)); structures 'gn:<head>' use oracle/p3d_gn.py.
# Gradients stored: up to 16 trainables of at most 40000 elements, spread over the creation order.
MORE = {
    "concat_b16_112": ("concat", 16, (1, 1, 2), (1, 16, 32, 32)),
    "unetpp_nonsa_b16_112": ("unet++nonsa", 16, (1, 1, 2), (1, 16, 32, 32)),
    "unetpp_ds_b16_112": ("unet++ds", 16, (1, 1, 2), (1, 16, 32, 32)),
    "gn_p3d_b16_112": ("gn:p3d", 16, (1, 1, 2), (1, 16, 32, 32)),
    "gn_decoder_b16_112": ("gn:decoder", 16, (1, 1, 2), (1, 16, 32, 32)),
}
GRADS = ["firstconv1", "conv3_0_1", "STB_1_2_T", "conv3_2_3", "dw3d_0", "batch_normalization_3/gamma",
         "conv3d_transpose/kernel", "conv3d_transpose_2/kernel", "deconv2_bn/beta", "conv3d/kernel",
         "conv3d_transpose_3/kernel", "conv3d_transpose_3/bias"]


def randomise_norm_params(params, seed=5):
    rng = np.random.default_rng(seed)
    for k, v in params.items():
        if k.endswith('gamma'):
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


def case_inputs(name):
    base, blocks, shape = CASES[name]
    cfg = p3d.NetConfig(base=base, blocks=blocks)
    params = randomise_norm_params(p3d.init_params(1, 'unet', cfg, dtype=np.float64))
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    return cfg, params, x, y


def compute(name, dtype=np.float64):
    cfg, params, x, y = case_inputs(name)
    params = {k: v.astype(dtype) for k, v in params.items()}
    pred_eval, _ = p3d.forward(params, x.astype(dtype), 0.0, False, 'unet', cfg, dtype)
    loss, pred_train, grads, _ = p3d.loss_and_grads(params, x.astype(dtype), y.astype(dtype), 0.0, True, 'unet', cfg, dtype)
    out = {"pred_eval": pred_eval.astype(np.float32), "pred_train": pred_train.astype(np.float32),
           "loss": np.float64(loss)}
    for g in GRADS:
        if grads[g].size <= 40000:          # keep the fixtures small
            out["grad:" + g] = grads[g].astype(np.float32)
    return out


def more_inputs(name):
    structure, base, blocks, shape = MORE[name]
    cfg = p3d.NetConfig(base=base, blocks=blocks)
    if structure.startswith("gn:"):
        params = p3d_gn.init_params(1, cfg, dtype=np.float64, head=structure[3:])
        rng = np.random.default_rng(7)
        for k, v in params.items():
            if k.endswith('gamma'):
                v[:] = rng.uniform(0.5, 1.5, v.shape)
            elif k.endswith(('beta', '/bias')):
                v[:] = rng.uniform(-0.3, 0.3, v.shape)
    else:
        params = randomise_norm_params(p3d.init_params(1, structure, cfg, dtype=np.float64))
        rng = np.random.default_rng(9)
        for k, v in params.items():
            if k.startswith('gamma'):           # attention mixing scalars: TF's initial 0 would switch the branch off
                v[:] = rng.uniform(0.4, 1.0, v.shape)
    x = p3d.synthetic_clip(0, shape + (3,))
    y = p3d.synthetic_target(3, shape)
    return structure, cfg, params, x, y


def compute_more(name, dtype=np.float64):
    structure, cfg, params, x, y = more_inputs(name)
    params = {k: v.astype(dtype) for k, v in params.items()}
    if structure.startswith("gn:"):
        head = structure[3:]
        pred_eval, _ = p3d_gn.forward(params, x.astype(dtype), 0.0, False, cfg, dtype, head=head)
        loss, pred_train, grads, _ = p3d_gn.loss_and_grads(params, x.astype(dtype), y.astype(dtype), 0.0, True, cfg, dtype, head=head)
    else:
        pred_eval, _ = p3d.forward(params, x.astype(dtype), 0.0, False, structure, cfg, dtype)
        loss, pred_train, grads, _ = p3d.loss_and_grads(params, x.astype(dtype), y.astype(dtype), 0.0, True, structure, cfg, dtype)
    out = {"pred_eval": pred_eval.astype(np.float32), "pred_train": pred_train.astype(np.float32), "loss": np.float64(loss)}
    small = [n for n, g in grads.items() if g.size <= 40000]
    for n in small[::max(1, len(small) // 16)][:16]:
        out["grad:" + n] = grads[n].astype(np.float32)
    return out


if __name__ == "__main__":
    for name in MORE:
        out = compute_more(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "loss", out["loss"], os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
    if "--all" not in sys.argv:
        sys.exit(0)             # the two p3d_unet fixtures are only rewritten on request
    for name in CASES:
        out = compute(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "loss", out["loss"], os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
