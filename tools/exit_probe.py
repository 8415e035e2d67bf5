import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = sys.argv[1]
if "torch" in mode:
    import torch
from sap3d_tensorflow_amd import P3DSession
s = P3DSession('unet', batch=1, frames=16, height=32, width=32, base=16, blocks=(1, 1, 1))
if "comm" in mode:
    s.comm_init(P3DSession.comm_unique_id())
import numpy as np
x = np.zeros((1, 16, 32, 32, 3), np.float32); y = np.zeros((1, 16, 32, 32), np.float32)
print(mode, "loss", s.train_step(x, y, 0.0))
if "noclose" not in mode:
    s.close()
print(mode, "exiting")
