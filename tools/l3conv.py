import sys, numpy as np
sys.path.insert(0, '.')
from sap3d_tensorflow_amd import ops
rng = np.random.default_rng(0)
x = rng.standard_normal((8, 2, 7, 7, 256)).astype(np.float32)
w = rng.standard_normal((1, 3, 3, 256, 256)).astype(np.float32)
for i in range(4): y = ops.conv3d(x, w, (1, 1, 1))
x = rng.standard_normal((8, 2, 7, 7, 1024)).astype(np.float32)
w = rng.standard_normal((1, 1, 1, 1024, 256)).astype(np.float32)
for i in range(4): y = ops.conv3d(x, w, (1, 1, 1))
