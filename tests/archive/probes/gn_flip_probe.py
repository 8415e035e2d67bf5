"""Diagnostic: how often, and where, do GN/CBAM gradients of the HIP path deviate from the fp64 oracle?
A CBAM arg-max flip in block k perturbs every gradient upstream of k and nothing downstream."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import p3d, p3d_gn
from test_gpu_net import GN_SMALL, _gn_params, make_session, rel_l2

cfg, shape = GN_SMALL[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
p64 = _gn_params(cfg, np.float64)
p32 = {k: v.astype(np.float32) for k, v in p64.items()}
x = p3d.synthetic_clip(0, shape + (3,)); y = p3d.synthetic_target(3, shape)
l64, pr64, g64, _ = p3d_gn.loss_and_grads(p64, x.astype(np.float64), y.astype(np.float64), 0.0, True, cfg, np.float64)
scale = np.median([np.linalg.norm(v) for v in g64.values()])
s = make_session(cfg, shape, p32, 'gn_p3d')
prev = None
for run in range(int(os.environ.get("PROBE_RUNS", "3"))):
    loss, pred = s.backward(x, y, 0.0)
    got = {n: s.get_grad(n) for n in g64}
    bad = [(n, rel_l2(got[n], g64[n], 1e-2 * scale)) for n in g64]
    worst = [(n, e) for n, e in bad if e > 1e-3]
    print("run", run, "loss", loss, "n>1e-3:", len(worst), "first", [(n, round(float(e), 5)) for n, e in worst[:2]], "last", [(n, round(float(e), 5)) for n, e in worst[-6:]])
    if run == 0 and worst:
        names = list(g64); last = max(names.index(n) for n, _ in worst)
        print("   boundary: last bad index", last, "of", len(names), "next good:", names[last + 1:last + 4])
    if prev is not None:
        d = max(rel_l2(got[n], prev[n], 1e-2 * scale) for n in g64)
        print("   run-to-run max rel-l2", d)
    prev = got
