import sys, numpy as np
sys.path.insert(0, '.')
from sap3d_tensorflow_amd import ops
rng = np.random.default_rng(0)
which = sys.argv[1] if len(sys.argv) > 1 else 'd3dgrad'
if which == 'd3dgrad':      # deconv3 input-gradient == strided 3x3x3 conv of dy_big
    x = rng.standard_normal((8, 8, 56, 56, 128)).astype(np.float32)
    w = rng.standard_normal((3, 3, 3, 128, 512)).astype(np.float32)
    for i in range(3): y = ops.conv3d(x, w, (2, 2, 2))
elif which == 'l1convS':
    x = rng.standard_normal((8, 8, 28, 28, 64)).astype(np.float32)
    w = rng.standard_normal((1, 3, 3, 64, 64)).astype(np.float32)
    for i in range(3): y = ops.conv3d(x, w, (1, 1, 1))
elif which == 'l1conv3':
    x = rng.standard_normal((8, 8, 28, 28, 64)).astype(np.float32)
    w = rng.standard_normal((1, 1, 1, 64, 256)).astype(np.float32)
    for i in range(3): y = ops.conv3d(x, w, (1, 1, 1))
print('done')
