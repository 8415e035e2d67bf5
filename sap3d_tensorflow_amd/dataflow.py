"""Frame pre-processing of the reference's loader (dataflow.py:187-216 `mapf`, gen_pred.py:117-121) on the GPU:
decoded uint8 frames in, float32 clip tensors out, one fused pass (channel flip, mean subtraction, cv2.INTER_LINEAR
resize, / 255) in csrc/metrics.hip.  Decoding (cv2.imread) and the tensorpack plumbing stay with the caller."""
import ctypes as C

import numpy as np

from ._lib import check, lib

# dataflow.py:194-196: mean_value = [98, 102, 90][::-1] -> per RGB channel
MEAN_RGB = (90.0, 102.0, 98.0)


def mapf_frames(frames_bgr, size=112, mean_rgb=MEAN_RGB, device=0):
    """[n, H0, W0, 3] uint8 BGR (what cv2.imread returns) -> [n, size, size, 3] float32, one clip of the x placeholder."""
    f = np.ascontiguousarray(frames_bgr, dtype=np.uint8)
    if f.ndim == 3:
        f = f[None]
    if f.ndim != 4 or f.shape[3] != 3 or f.size == 0:
        raise ValueError("expected [n, H, W, 3] uint8 frames")
    H, W = (size, size) if np.isscalar(size) else size
    out = np.empty((f.shape[0], H, W, 3), np.float32)
    mean = (C.c_float * 3)(*[float(v) for v in mean_rgb])
    check(lib().p3d_mapf_frames(device, f.ctypes.data_as(C.POINTER(C.c_ubyte)), f.shape[0], f.shape[1], f.shape[2], mean, H, W,
                                out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def mapf_density(maps_grey, size=112, device=0):
    """[n, H0, W0] uint8 density maps (cv2.IMREAD_GRAYSCALE) -> [n, size, size] float32 in [0, 1], the y placeholder."""
    f = np.ascontiguousarray(maps_grey, dtype=np.uint8)
    if f.ndim == 2:
        f = f[None]
    if f.ndim != 3 or f.size == 0:
        raise ValueError("expected [n, H, W] uint8 maps")
    H, W = (size, size) if np.isscalar(size) else size
    out = np.empty((f.shape[0], H, W), np.float32)
    check(lib().p3d_mapf_density(device, f.ctypes.data_as(C.POINTER(C.c_ubyte)), f.shape[0], f.shape[1], f.shape[2], H, W,
                                 out.ctypes.data_as(C.POINTER(C.c_float))))
    return out
