"""Diagnostic: op-level conv parity at in-network extents (larger M than tests/test_gpu_ops.py uses)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import nn
from sap3d_tensorflow_amd import ops

rng = np.random.default_rng(0)
def rnd(shape): return rng.standard_normal(shape).astype(np.float32)
def err(got, want): return np.abs(got.astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-30)
for xs, k, co, s in [((2, 16, 32, 32, 8), (3, 3, 3), 4, (1, 1, 1)), ((2, 8, 16, 16, 32), (3, 3, 3), 8, (1, 1, 1)),
                     ((2, 16, 32, 32, 8), (3, 3, 3), 8, (2, 2, 2)), ((2, 8, 16, 16, 32), (3, 3, 3), 32, (1, 1, 1))]:
    x = rnd(xs); w = rnd(k + (xs[4], co)) * 0.1; b = rnd((co,))
    want = nn.conv3d_forward(x.astype(np.float64), w.astype(np.float64), s) + b
    got = ops.conv3d(x, w, s, bias=b)
    dy = rnd(want.shape)
    want_dw = nn.conv3d_backward_filter(x.astype(np.float64), dy.astype(np.float64), w.shape, s)
    got_dw, got_db = ops.conv3d_backprop_filter(x, w.shape, dy, s, with_bias=True)
    want_dx = nn.conv3d_backward_input(dy.astype(np.float64), w.astype(np.float64), s, xs)
    got_dx = ops.conv3d_backprop_input(xs, w, dy, s)
    print(xs, k, co, s, "fwd %.2e dw %.2e db %.2e dx %.2e" % (err(got, want), err(got_dw, want_dw),
          err(got_db, dy.astype(np.float64).reshape(-1, co).sum(0)), err(got_dx, want_dx)))
    # transposed conv with the same geometry: y = deconv(x') where x' has the conv's output shape
    kern = rnd(k + (xs[4], co)) * 0.1
    t = nn.Tape()
    xin = rnd(want.shape)
    if all(want.shape[1 + a] * s[a] == xs[1 + a] for a in range(3)):
        wt = nn.conv3d_transpose(t, nn.Var(xin.astype(np.float64)), nn.Var(kern.astype(np.float64)), s, nn.Var(np.zeros(xs[4]))).data
        gt = ops.conv3d_transpose(xin, kern, s, bias=np.zeros(xs[4], np.float32))
        print("    deconv fwd %.2e" % err(gt, wt))
