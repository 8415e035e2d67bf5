"""ORACLE (test infrastructure only -- never imported by the product): numpy restatement of the frame pre-processing of
/root/reference/dataflow.py:187-216 (`mapf`): BGR uint8 frame -> RGB -> minus the channel means -> tensorpack
imgaug.Resize(112) (= cv2.resize, INTER_LINEAR) -> / 255; density map: grey uint8 -> resize -> / 255.

PARITY UNPINNED: cv2 and tensorpack are not installed, so the resize is restated from OpenCV's documented INTER_LINEAR
rule for float32 images (source coordinate (d + 0.5) * scale - 0.5 computed in double and cast to float, floor, weight 0
at a clamped border, horizontal then vertical pass in float32) and nothing here was checked against cv2 itself.
"""
import numpy as np

MEAN_RGB = np.array([98, 102, 90], dtype=np.float32)[::-1].copy()      # dataflow.py:194-196: [90, 102, 98]


def _coef(dst, src):
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    lo = sx < 0
    sx[lo] = 0; fx[lo] = 0
    hi = sx >= src - 1
    sx[hi] = src - 1; fx[hi] = 0
    s1 = np.minimum(sx + 1, src - 1)
    return sx, s1, fx


def resize_linear(im, H, W):
    """cv2.resize(im, (W, H), interpolation=cv2.INTER_LINEAR) for a float32 image [H0, W0] or [H0, W0, C]."""
    im = np.asarray(im, dtype=np.float32)
    x0, x1, wx = _coef(W, im.shape[1])
    y0, y1, wy = _coef(H, im.shape[0])
    if im.ndim == 3:
        wx = wx[None, :, None]; wyb = wy[:, None, None]
    else:
        wx = wx[None, :]; wyb = wy[:, None]
    one = np.float32(1)
    rows = (im[:, x0] * (one - wx)).astype(np.float32) + (im[:, x1] * wx).astype(np.float32)
    out = (rows[y0] * (one - wyb)).astype(np.float32) + (rows[y1] * wyb).astype(np.float32)
    return out.astype(np.float32)


def mapf_frame(bgr_u8, H=112, W=112):
    """dataflow.py:202-208 for one decoded frame."""
    im = np.asarray(bgr_u8)[:, :, ::-1].astype(np.float32) - MEAN_RGB[None, None, :]
    return (resize_linear(im, H, W) / np.float32(255.0)).astype(np.float32)


def mapf_density(grey_u8, H=112, W=112):
    """dataflow.py:210-214 for one decoded density map."""
    return (resize_linear(np.asarray(grey_u8).astype(np.float32), H, W) / np.float32(255.0)).astype(np.float32)
