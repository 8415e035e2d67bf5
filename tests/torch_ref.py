"""The torch-CPU composition lives in oracle/torch_port.py (bench.py's cpu_baseline may only import from oracle/)."""
from oracle.torch_port import *          # noqa: F401,F403
from oracle.torch_port import TorchP3D, TorchP3DGN, smooth_l1_sum, conv3d_same, conv3d_transpose_same, max_pool_same   # noqa: F401
