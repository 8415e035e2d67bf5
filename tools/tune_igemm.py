import sys, os, numpy as np
sys.path.insert(0, '.')
from sap3d_tensorflow_amd import ops
rng = np.random.default_rng(0)
B = 8
cases = {
 'L3conv1': ((B,2,7,7,1024),(1,1,1),256),
 'L3convS': ((B,2,7,7,256),(1,3,3),256),
 'L3convT': ((B,2,7,7,256),(3,1,1),256),
 'L3conv3': ((B,2,7,7,256),(1,1,1),1024),
 'L2convS': ((B,4,14,14,128),(1,3,3),128),
 'L2conv1': ((B,4,14,14,512),(1,1,1),128),
}
for name,(xs,k,co) in cases.items():
    x = rng.standard_normal(xs).astype(np.float32)
    w = rng.standard_normal(k+(xs[4],co)).astype(np.float32)
    for i in range(3):
        y = ops.conv3d(x, w, (1,1,1))
print('done')
