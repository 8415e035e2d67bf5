"""Diagnostic: forward error of the HIP path vs the fp64 oracle, next to the fp32 oracle's own error, per tap."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import p3d, p3d_gn
from test_gpu_net import GN_DECODER, _gn_params, make_session

cfg, shape, _ = GN_DECODER[0]
p64 = _gn_params(cfg, np.float64, 'decoder')
p32 = {k: v.astype(np.float32) for k, v in p64.items()}
x = p3d.synthetic_clip(0, shape + (3,))
s = make_session(cfg, shape, p32, 'gn_p3d_decoder')
w64, g64 = p3d_gn.forward(p64, x.astype(np.float64), 0.0, False, cfg, np.float64, head='decoder')
w32, g32 = p3d_gn.forward(p32, x, 0.0, False, cfg, np.float32, head='decoder')
got = s.forward(x, 0.0, False)
for name in ['deconv_pool2', 'deconv_pool3', 'deconv_pool4', 'conv_concat', 'decoder1_conv1', 'decoder1_deconv',
             'decoder1_conv2', 'decoder2_conv1', 'decoder2_deconv', 'decoder2_conv2']:
    w = g64.tape.taps[name].data
    a = s.activation(name); o = g32.tape.taps[name].data
    m = np.abs(w).max()
    near = lambda v: int(((v > 0) != (w > 0)).sum())
    print("%-16s max|w| %.3g  hip err %.2e (rms %.2e) relu-sign flips %d | fp32 oracle err %.2e (rms %.2e) flips %d | n=%d" % (
        name, m, np.abs(a - w).max() / m, np.sqrt(((a - w) ** 2).mean()) / m, near(a),
        np.abs(o - w).max() / m, np.sqrt(((o - w) ** 2).mean()) / m, near(o), w.size))
print("pred hip err %.2e fp32 %.2e" % (np.abs(got - w64).max() / np.abs(w64).max(), np.abs(w32 - w64).max() / np.abs(w64).max()))
