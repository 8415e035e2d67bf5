import sys, faulthandler, numpy as np
faulthandler.dump_traceback_later(90, exit=True)
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import p3d
from test_gpu_net import make_session, randomise_norm_params, SMALL
cfg, shape = SMALL[0]
params = randomise_norm_params(p3d.init_params(1,'unet',cfg))
print('creating', flush=True)
s = make_session(cfg, shape, params)
print('created', flush=True)
x = p3d.synthetic_clip(0, shape+(3,))
got = s.forward(x, 0.0, False)
print('forward done', got.mean(), flush=True)
y = p3d.synthetic_target(3, shape)
loss, pred = s.backward(x, y)
print('backward done', loss, flush=True)
