"""Diagnostic: per-tensor gradient error of the unet++ds (self attention) structure vs the fp64 oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import p3d
from test_gpu_net import DS_CASES, make_session, rel_l2, randomise_norm_params

st = 'unet++ds'
cfg, shape = DS_CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
p64 = randomise_norm_params(p3d.init_params(1, st, cfg, dtype=np.float64))
p32 = {k: v.astype(np.float32) for k, v in p64.items()}
x = p3d.synthetic_clip(0, shape + (3,)); y = p3d.synthetic_target(3, shape)
s = make_session(cfg, shape, p32, st)
l64, pr64, g64, _ = p3d.loss_and_grads(p64, x.astype(np.float64), y.astype(np.float64), 0.0, True, st, cfg, np.float64)
_, _, g32, _ = p3d.loss_and_grads(dict(p32), x, y, 0.0, True, st, cfg, np.float32)
loss, pred = s.backward(x, y, 0.0)
print("loss", loss, l64, "pred err", np.abs(pred - pr64).max())
names = list(g64)
first_sa = min(i for i, n in enumerate(names) if '_sa' in n)
for n in names[first_sa - 4:]:
    w = g64[n]; nw = np.linalg.norm(w)
    e = np.linalg.norm(s.get_grad(n) - w) / max(nw, 1e-30)
    e32 = np.linalg.norm(g32[n] - w) / max(nw, 1e-30)
    print("%-34s |g| %.3e  hip relerr %.2e  fp32-oracle relerr %.2e %s" % (n, nw, e, e32, "<<<" if e > 20 * e32 + 1e-3 else ""))
