"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the float64 oracle).
CPU: the float64 and float32 oracles still reproduce them.  GPU: libp3dhip reproduces them."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as mg      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def rel_l2(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30)


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_oracle_reproduces_golden(name):
    gold = load(name)
    out = mg.compute(name, np.float64)
    assert abs(out["loss"] - gold["loss"]) <= 1e-9 * abs(gold["loss"])
    for k in gold.files:
        if k != "loss":
            assert np.array_equal(out[k], gold[k]), k
    out32 = mg.compute(name, np.float32)
    assert abs(out32["loss"] - gold["loss"]) <= 1e-6 * abs(gold["loss"])
    assert np.abs(out32["pred_eval"] - gold["pred_eval"]).max() < 1e-4
    assert np.abs(out32["pred_train"] - gold["pred_train"]).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_hip_reproduces_golden(name):
    from sap3d_tensorflow_amd import P3DSession
    gold = load(name)
    cfg, params, x, y = mg.case_inputs(name)
    B, T, H, W = mg.CASES[name][2]
    s = P3DSession('unet', batch=B, frames=T, height=H, width=W, base=cfg.base, blocks=cfg.blocks)
    s.load({k: v.astype(np.float32) for k, v in params.items()})
    # saliency maps within 1e-3 relative (north star); they are sigmoid outputs near 0.5
    pe = s.forward(x, 0.0, False)
    assert (np.abs(pe - gold["pred_eval"]) / np.abs(gold["pred_eval"])).max() < 1e-3
    loss, pt = s.backward(x, y, 0.0)
    assert (np.abs(pt - gold["pred_train"]) / np.abs(gold["pred_train"])).max() < 1e-3
    assert abs(loss - gold["loss"]) <= 1e-5 * abs(gold["loss"])
    # gradients: fp32 noise floor of this net is ~1e-2 rel-L2 (tests/test_oracle_vs_torch.py)
    errs = {g: rel_l2(s.get_grad(g), gold["grad:" + g]) for g in mg.GRADS
            if "grad:" + g in gold.files and np.linalg.norm(gold["grad:" + g]) > 1e-3}
    assert max(errs.values()) < 6e-2, errs
    assert np.median(list(errs.values())) < 2e-2, errs
    s.close()
