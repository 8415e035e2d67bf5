"""Diagnostic: does a backward pass depend on what the previous step left in device memory?
Alternates two inputs A/B; any read-before-write of scratch shows up as an error right after a switch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import p3d, p3d_gn
from test_gpu_net import GN_SMALL, SMALL, _gn_params, make_session, rel_l2, randomise_norm_params

which = sys.argv[1] if len(sys.argv) > 1 else 'gn_p3d'
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if which == 'gn_p3d':
    cfg, shape = GN_SMALL[idx]
    p64 = _gn_params(cfg, np.float64)
    lg = lambda x, y: p3d_gn.loss_and_grads(p64, x.astype(np.float64), y.astype(np.float64), 0.0, True, cfg, np.float64)
else:
    cfg, shape = SMALL[idx]
    p64 = randomise_norm_params(p3d.init_params(1, which, cfg, dtype=np.float64))
    lg = lambda x, y: p3d.loss_and_grads(p64, x.astype(np.float64), y.astype(np.float64), 0.0, True, which, cfg, np.float64)
p32 = {k: v.astype(np.float32) for k, v in p64.items()}
data = {}
for tag, (sx, sy) in {'A': (0, 3), 'B': (5, 9)}.items():
    x = p3d.synthetic_clip(sx, shape + (3,)); y = p3d.synthetic_target(sy, shape)
    l, pr, g, _ = lg(x, y)
    data[tag] = (x, y, l, g, np.median([np.linalg.norm(v) for v in g.values()]))
s = make_session(cfg, shape, p32, which)
names = list(data['A'][3])
for run, tag in enumerate("AABBABBA"):
    x, y, l, g, scale = data[tag]
    loss, pred = s.backward(x, y, 0.0)
    errs = [(n, rel_l2(s.get_grad(n), g[n], 1e-2 * scale)) for n in names]
    bad = [(n, round(float(e), 5)) for n, e in errs if e > 1e-3]
    print("run", run, tag, "loss rel err %.2e" % (abs(loss - l) / abs(l)), "bad:", len(bad), "last bad:", bad[-3:] if bad else "")
s.close()
