"""ORACLE (test infrastructure only -- never imported by the product): numpy restatement of the frame pre-processing of
/root/reference/dataflow.py:187-216 (`mapf`): BGR uint8 frame -> RGB -> minus the channel means -> tensorpack
imgaug.Resize(112) (= cv2.resize, INTER_LINEAR) -> / 255; density map: grey uint8 -> resize -> / 255.

PARITY UNPINNED: cv2 and tensorpack are not installed, so both resizes are restated from OpenCV's published algorithm
(modules/imgproc/src/resize.cpp: resizeGeneric_ with HResizeLinear / VResizeLinear) and nothing here was checked against
cv2 itself; a hand-computed fixture in tests/test_metrics.py pins the arithmetic below against later edits.
 * float32 images (the RGB frames, float after the mean subtraction): source coordinate (d + 0.5) * scale - 0.5 computed in
   double and cast to float, floor, weight 0 at a clamped border, horizontal then vertical pass in float32;
 * uint8 images (the grey density maps, dataflow.py:210-214 resize them BEFORE the division): the fixed-point path --
   coefficients cvRound(w * 2048) as shorts, horizontal pass into int32, vertical pass
   uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2) -- so the result is a uint8 image and y = k / 255.
   (OpenCV builds that route 8-bit INTER_LINEAR through IPP or OpenCL may differ from this generic path by one level.)
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


def _coef_u8(dst, src, clamp_weight):
    """cv::resize's 8-bit coefficient tables for one axis: source index pair and the two 11-bit weights.
    clamp_weight: the horizontal tables zero the weight at a clamped border; the vertical pass clips the ROW INDICES
    instead and keeps the weights (resize.cpp: xofs / ialpha vs. the srows clip in resizeGeneric_Invoker)."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s0 = np.floor(f).astype(np.int64)
    f = (f - s0.astype(np.float32)).astype(np.float32)
    if clamp_weight:
        lo = s0 < 0
        s0[lo] = 0; f[lo] = 0
        hi = s0 >= src - 1
        s0[hi] = src - 1; f[hi] = 0
    w0 = np.rint((np.float32(1) - f).astype(np.float32) * np.float32(2048)).astype(np.int64)      # saturate_cast<short>(cvRound)
    w1 = np.rint(f * np.float32(2048)).astype(np.int64)
    i0 = np.clip(s0, 0, src - 1)
    i1 = np.clip(s0 + 1, 0, src - 1)
    return i0, i1, w0, w1


def resize_linear_u8(im, H, W):
    """cv2.resize(im, (W, H), interpolation=cv2.INTER_LINEAR) for a uint8 image [H0, W0]: OpenCV's fixed-point path."""
    im = np.asarray(im)
    if im.dtype != np.uint8 or im.ndim != 2:
        raise ValueError("expected a 2-D uint8 image")
    if im.shape == (H, W):
        return im.copy()
    x0, x1, a0, a1 = _coef_u8(W, im.shape[1], True)
    y0, y1, b0, b1 = _coef_u8(H, im.shape[0], False)
    src = im.astype(np.int64)
    rows = src[:, x0] * a0[None, :] + src[:, x1] * a1[None, :]                     # HResizeLinear: int32 in OpenCV, exact here
    s0, s1 = rows[y0], rows[y1]
    out = (((b0[:, None] * (s0 >> 4)) >> 16) + ((b1[:, None] * (s1 >> 4)) >> 16) + 2) >> 2     # VResizeLinear<uchar, int, short, ...>
    return np.clip(out, 0, 255).astype(np.uint8)


def mapf_density(grey_u8, H=112, W=112):
    """dataflow.py:210-214 for one decoded density map: uint8 resize, then / 255. (float64 in numpy; fed as float32)."""
    return (resize_linear_u8(np.asarray(grey_u8, dtype=np.uint8), H, W) / 255.0).astype(np.float32)
