import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from sap3d_tensorflow_amd import P3DSession, synthetic
x = synthetic.synthetic_clip(0, (2, 16, 112, 112, 3)); y = synthetic.synthetic_target(3, (2,16,112,112))
s = P3DSession('unet', batch=2, seed=1)
full = s.forward(x, 0.0, True); l32,_ = s.backward(x,y,0.0)
s.set_pointwise_fp16(True)
half = s.forward(x, 0.0, True); l16,_ = s.backward(x,y,0.0)
rel = np.abs(half-full)/np.abs(full)
print("pred range", full.min(), full.max(), "rel max %.3g mean %.3g q99 %.3g q999 %.3g"%(rel.max(), rel.mean(), np.quantile(rel,0.99), np.quantile(rel,0.999)), "abs max", np.abs(half-full).max(), "loss", l32, l16, abs(l16-l32)/l32)
for name in ['conv1_custom_bn_relu','block0/out','block2/out','block10/out','block20/out','block46/out','deconv3_re','deconv4_conv1','logits']:
    s.set_pointwise_fp16(False); s.forward(x,0.0,True); a=s.activation(name)
    s.set_pointwise_fp16(True); s.forward(x,0.0,True); b=s.activation(name)
    print("%-22s rel-L2 %.3e  max|a| %.3g max abs diff %.3g"%(name, np.linalg.norm(a-b)/np.linalg.norm(a), np.abs(a).max(), np.abs(a-b).max()))
