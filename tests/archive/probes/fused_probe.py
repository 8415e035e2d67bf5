"""Fused vs unfused vs oracle loss trajectories (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import p3d
from tests.test_gpu_fused import CASES, randomise, session
cfg, shape = CASES[0]
p64 = randomise(p3d.init_params(1, 'unet', cfg, dtype=np.float64))
p32 = {k: v.astype(np.float32) for k, v in p64.items()}
x = p3d.synthetic_clip(0, shape + (3,))
y = p3d.synthetic_target(3, shape)
for mode in (0, 1):
    s = session(cfg, shape, p32)
    s.set_bn_fusion(mode)
    s.set_adam(1e-3)
    print(mode, [s.train_step(x, y, dropout=0.0) for _ in range(4)])
    s.close()
state = {'t': 0, 'm': {}, 'v': {}}
print('oracle64', [p3d.train_step(p64, state, x.astype(np.float64), y.astype(np.float64), lr=1e-3, cfg=cfg, dtype=np.float64)[0] for _ in range(4)])
state = {'t': 0, 'm': {}, 'v': {}}
p32b = {k: v.copy() for k, v in p32.items()}
print('oracle32', [p3d.train_step(p32b, state, x, y, lr=1e-3, cfg=cfg, dtype=np.float32)[0] for _ in range(4)])
