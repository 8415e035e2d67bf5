"""Diagnostic: directional-derivative check per tensor (which gradients predict the loss change they should?)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sap3d_tensorflow_amd import P3DSession, synthetic

B, T, H, W = 2, 16, 112, 112
s = P3DSession('unet', batch=B, frames=T, height=H, width=W, seed=3)
x = synthetic.synthetic_clip(0, (B, T, H, W, 3)); y = synthetic.synthetic_target(3, (B, T, H, W))
theta = s.save()
rng = np.random.default_rng(0)
for n in sorted(theta):
    if n.endswith('/beta'):
        theta[n] = rng.uniform(-0.2, 0.2, theta[n].shape).astype(np.float32)
s.load(theta)
loss0, _ = s.backward(x, y, 0.0)
l1, _ = s.backward(x, y, 0.0)
print("loss", loss0, "repeat", l1)
trainable = [n for n, _, t in s.variables() if t]
g = {n: s.get_grad(n).astype(np.float64) for n in trainable}
norms = sorted(((float((v * v).sum()), n) for n, v in g.items()), reverse=True)
tot = sum(v for v, _ in norms)
print("gnorm", np.sqrt(tot))
for v, n in norms[:12]:
    print("  %-40s |g|^2 share %.4f  |g| %.4g  shape %s" % (n, v / tot, np.sqrt(v), g[n].shape))
def check(names, rel=3e-3):
    gn = np.sqrt(sum((g[n] ** 2).sum() for n in names))
    eps = rel * abs(loss0) / gn
    Ls, mv = [], []
    for sign in (1.0, -1.0):
        m = dict(theta)
        for n in names:
            m[n] = (theta[n].astype(np.float64) + sign * eps * g[n] / gn).astype(np.float32)
        s.load(m); Ls.append(s.backward(x, y, 0.0)[0]); mv.append(m)
    pred = sum(float((g[n] * (mv[0][n].astype(np.float64) - mv[1][n].astype(np.float64))).sum()) for n in names)
    return Ls[0] - Ls[1], pred, eps
for v, n in norms[:6]:
    m, p, e = check([n])
    print("single %-40s measured %.4f predicted %.4f eps %.3g" % (n, m, p, e))
for grp, sel in [("all", trainable), ("kernels only", [n for n in trainable if not n.endswith(('gamma', 'beta', 'bias'))]),
                 ("gamma/beta only", [n for n in trainable if n.endswith(('gamma', 'beta'))]),
                 ("biases only", [n for n in trainable if n.endswith('bias')])]:
    for rel in (3e-3, 3e-2):
        m, p, e = check(sel, rel)
        print("group %-16s rel %.0e measured %.4f predicted %.4f eps %.3g" % (grp, rel, m, p, e))

print("--- L(t) along single-tensor gradient directions, t in units of eps0 = 3e-3*L/|g_n|")
for n in ['firstconv1', 'conv3_1_1', 'conv3d_transpose_2/kernel', 'conv3_40_1', 'batch_normalization_100/gamma']:
    if n not in g:
        continue
    gn = np.sqrt((g[n] ** 2).sum()); eps0 = 3e-3 * abs(loss0) / gn
    out = []
    for mult in (1 / 256, 1 / 64, 1 / 16, 1 / 4, 1, 4):
        Ls, mv = [], []
        for sign in (1.0, -1.0):
            m = dict(theta); m[n] = (theta[n].astype(np.float64) + sign * mult * eps0 * g[n] / gn).astype(np.float32)
            s.load(m); Ls.append(s.backward(x, y, 0.0)[0]); mv.append(m[n])
        pred = float((g[n] * (mv[0].astype(np.float64) - mv[1].astype(np.float64))).sum())
        out.append("t=%.4g: meas %.3f pred %.3f" % (mult, Ls[0] - Ls[1], pred))
    print(n, "|g|", gn, " | ".join(out))
