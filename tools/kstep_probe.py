"""Diagnostic: igemm2<64,64> time vs number of K-steps at stage-3 size (M = 784 rows): slope = cost of one pipeline step,
intercept = fixed cost of the launch.  Run under rocprofv3 --kernel-trace."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sap3d_tensorflow_amd import ops
rng = np.random.default_rng(0)
os.environ["P3D_SPLITS"] = "1"
for K in (32, 64, 128, 256, 512, 1024, 2048):
    x = rng.standard_normal((8, 2, 7, 7, K)).astype(np.float32)
    w = (rng.standard_normal((1, 1, 1, K, 1024)) * 0.05).astype(np.float32)
    for _ in range(4):
        ops.conv3d(x, w, (1, 1, 1))
print("done")
