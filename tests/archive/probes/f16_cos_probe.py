"""Cosine of the fp16-pointwise gradients against the float64 oracle, fused and unfused launch lists (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import p3d
from tests.test_gpu_net import SMALL, randomise_norm_params, make_session
cfg, shape = SMALL[1]
p64 = randomise_norm_params(p3d.init_params(1, 'unet', cfg, dtype=np.float64))
p32 = {k: v.astype(np.float32) for k, v in p64.items()}
x = p3d.synthetic_clip(0, shape + (3,))
y = p3d.synthetic_target(3, shape)
l64, pr64, g64, _ = p3d.loss_and_grads(p64, x.astype(np.float64), y.astype(np.float64), 0.0, True, 'unet', cfg, np.float64)
for mode in (0, 1):
    s = make_session(cfg, shape, p32)
    s.set_bn_fusion(mode)
    s.set_pointwise_fp16(True)
    s.backward(x, y, 0.0)
    cs = []
    for n, w in g64.items():
        if np.linalg.norm(w) > 1e-6 * max(np.abs(l64), 1.0) and not n.endswith('bias'):
            got = s.get_grad(n).astype(np.float64)
            cs.append((float((got * w).sum() / (np.linalg.norm(got) * np.linalg.norm(w))), n))
    cs.sort()
    print(mode, cs[:6])
    s.close()
