"""Synthetic clips with the value law of the reference's loader (dataflow.py:204-208, gen_pred.py:117-121):
RGB uint8 minus the channel means [90,102,98], divided by 255; targets uniform in [0,1).
Seeded with numpy's PCG64 so every box draws identical inputs."""
import numpy as np

MEAN_RGB = np.array([90, 102, 98], np.float32)


def synthetic_clip(seed, shape):
    """shape [B,T,H,W,3] float32 in about [-0.40, 0.65]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u8 = rng.integers(0, 256, size=shape).astype(np.float32)
    return ((u8 - MEAN_RGB) / 255.0).astype(np.float32)


def synthetic_target(seed, shape):
    """shape [B,T,H,W] float32 saliency targets in [0,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random(size=shape, dtype=np.float32)
