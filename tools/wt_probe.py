"""Diagnostic: same-FLOP forward conv (weights [K][N], WT=0) vs input gradient (weights read transposed, WT=1) on
igemm2 -- is the b128 B-fragment path faster?  Run under rocprofv3 --kernel-trace --stats."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sap3d_tensorflow_amd import ops
rng = np.random.default_rng(0)
for (xs, k, co) in [((8, 4, 28, 28, 512), (3, 3, 3), 512), ((8, 8, 28, 28, 256), (1, 1, 1), 256), ((8, 4, 28, 28, 128), (1, 3, 3), 128)]:
    x = rng.standard_normal(xs).astype(np.float32); w = (rng.standard_normal(k + (xs[4], co)) * 0.05).astype(np.float32)
    for _ in range(3):
        y = ops.conv3d(x, w, (1, 1, 1))
        dx = ops.conv3d_backprop_input(xs, w, y, (1, 1, 1))
print("done")
