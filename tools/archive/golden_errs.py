"""Measurement behind the gradient gates of tests/test_golden.py: rel-L2 error of every HIP gradient against the float64
golden fixtures (max / median per fixture).  Run on the GPU box; prints one JSON line per fixture."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden import make_golden as mg
from sap3d_tensorflow_amd import P3DSession

def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))

STRUCTURE_OF = {"gn:p3d": "gn_p3d", "gn:decoder": "gn_p3d_decoder"}
for name in sorted(mg.CASES):
    gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    cfg, params, x, y = mg.case_inputs(name)
    B, T, H, W = mg.CASES[name][2]
    s = P3DSession("unet", batch=B, frames=T, height=H, width=W, base=cfg.base, blocks=cfg.blocks)
    s.load({k: v.astype(np.float32) for k, v in params.items()})
    loss, _ = s.backward(x, y, 0.0)
    errs = sorted(rel_l2(s.get_grad(g), gold["grad:" + g]) for g in mg.GRADS if "grad:" + g in gold.files and np.linalg.norm(gold["grad:" + g]) > 1e-3)
    print(json.dumps(dict(fixture=name, n=len(errs), max=errs[-1], median=errs[len(errs) // 2], p90=errs[int(0.9 * len(errs))])), flush=True)
    s.close()
for name in sorted(mg.MORE):
    gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    structure, cfg, params, x, y = mg.more_inputs(name)
    B, T, H, W = mg.MORE[name][3]
    s = P3DSession(STRUCTURE_OF.get(structure, structure), batch=B, frames=T, height=H, width=W, base=cfg.base, blocks=cfg.blocks)
    s.load({k: v.astype(np.float32) for k, v in params.items()})
    loss, _ = s.backward(x, y, 0.0)
    errs = sorted(rel_l2(s.get_grad(k[5:]), gold[k]) for k in gold.files if k.startswith("grad:") and np.linalg.norm(gold[k]) > 1e-3)
    print(json.dumps(dict(fixture=name, n=len(errs), max=errs[-1], median=errs[len(errs) // 2], p90=errs[int(0.9 * len(errs))])), flush=True)
    s.close()
