"""Diagnostic: per-tensor gradient error of the GN decoder-block head against the fp64 oracle (and the fp32 oracle's own)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import p3d, p3d_gn
from test_gpu_net import GN_DECODER, _gn_params, make_session, rel_l2

cfg, shape, _ = GN_DECODER[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
p64 = _gn_params(cfg, np.float64, 'decoder')
p32 = {k: v.astype(np.float32) for k, v in p64.items()}
x = p3d.synthetic_clip(0, shape + (3,)); y = p3d.synthetic_target(3, shape)
s = make_session(cfg, shape, p32, 'gn_p3d_decoder')
keep = None
if drop > 0:
    s.forward(x, 0.0, False)
    base = s.activation('decoder2_conv2')
    s.backward(x, y, dropout=drop, seed=11)
    keep = np.where(base != 0, s.activation('decoder2_conv2') != 0, True)
k64 = None if keep is None else keep.astype(np.float64)
k32 = None if keep is None else keep.astype(np.float32)
l64, pr64, g64, _ = p3d_gn.loss_and_grads(p64, x.astype(np.float64), y.astype(np.float64), drop, True, cfg, np.float64, head='decoder', keep_mask=k64)
l32, _, g32, _ = p3d_gn.loss_and_grads(dict(p32), x, y, drop, True, cfg, np.float32, head='decoder', keep_mask=k32)
scale = np.median([np.linalg.norm(v) for v in g64.values()])
for run in range(3):
    loss, pred = s.backward(x, y, dropout=drop, seed=11)
    print("run", run, "loss", loss, l64, "pred err", np.abs(pred - pr64).max())
    for n in g64:
        e = rel_l2(s.get_grad(n), g64[n], 1e-2 * scale)
        e32 = rel_l2(g32[n], g64[n], 1e-2 * scale)
        if e > 1e-3 or run == 0 and 'cbam_0' in n:
            print("   %-44s hip %.5f  fp32-oracle %.5f  |g| %.3g" % (n, e, e32, np.linalg.norm(g64[n])))
