"""the validation metrics (utils/metrics.py) and the frame pre-processing (dataflow.py
mapf).  CPU part: the numpy oracle against closed forms and against the reference's literal loops.  GPU part: the HIP
kernels (through the C ABI) against the oracle."""
import numpy as np
import pytest

from oracle import dataflow as odf
from oracle import metrics as om


def _maps(seed, n=1, shape=(112, 112), fix_rate=0.02):
    rng = np.random.default_rng(seed)
    s = rng.random((n,) + shape).astype(np.float32)
    t = rng.random((n,) + shape).astype(np.float32)
    f = (rng.random((n,) + shape) < fix_rate).astype(np.float32)
    return s, t, f


# ---- oracle pinned by closed forms ---------------------------------------------------------------------------------
def test_oracle_closed_forms():
    s, t, f = _maps(0)
    s, t, f = s[0], t[0], f[0]
    assert om.CC(s, s) == pytest.approx(1.0, abs=1e-12)
    assert om.CC(s, -3 * s + 2) == pytest.approx(-1.0, abs=1e-12)
    assert om.SIM(s, s) == pytest.approx(1.0, abs=1e-12)
    assert 0 < om.SIM(s, t) < 1
    # NSS: one fixated pixel -> its z-score
    one = np.zeros_like(s); one[5, 7] = 1
    assert om.NSS(s, one) == pytest.approx((s[5, 7] - s.astype(np.float64).mean()) / s.astype(np.float64).std(), rel=1e-12)
    # AUC: a map that ranks every fixation above every other pixel scores 1, the reversed ranking 0
    perfect = np.where(f > 0.5, 1 + s, 0.5 * s).astype(np.float32)      # distinct values: equal thresholds would tie
    assert om.AUC_Judd(perfect, f) == pytest.approx(1.0, abs=1e-12)
    assert om.AUC_Judd(-perfect, f) < 0.01              # (the trapezoid between the last threshold and (1,1) leaves a sliver)
    assert np.isnan(om.AUC_Judd(s, np.zeros_like(f)))
    assert np.isnan(om.NSS(s, np.zeros_like(f)))
    assert np.isnan(om.CC(np.ones_like(s), s))          # flat map: 0/0, as in the reference


def test_oracle_auc_judd_equals_the_reference_loop():
    """utils/metrics.py:76-85 literally (np.sum(S >= thresh) per threshold) on small maps, ties included."""
    rng = np.random.default_rng(3)
    for quant in (None, 8):
        s = rng.random((24, 20)).astype(np.float32)
        if quant:
            s = np.floor(s * quant).astype(np.float32) / quant        # many equal values
        f = (rng.random((24, 20)) < 0.1).astype(np.float32)
        S = s.ravel().astype(np.float64); F = f.ravel() > 0.5
        S_fix = S[F]; n_fix = len(S_fix); n_pixels = len(S)
        thresholds = sorted(S_fix, reverse=True)
        tp = np.zeros(len(thresholds) + 2); fp = np.zeros(len(thresholds) + 2)
        tp[-1] = 1; fp[-1] = 1
        for k, thresh in enumerate(thresholds):
            above_th = np.sum(S >= thresh)
            tp[k + 1] = (k + 1) / float(n_fix)
            fp[k + 1] = (above_th - k - 1) / float(n_pixels - n_fix)
        want = (getattr(np, "trapezoid", None) or np.trapz)(tp, fp)
        assert om.AUC_Judd(s, f) == pytest.approx(want, abs=1e-14)


def test_oracle_resize_rules():
    rng = np.random.default_rng(1)
    im = rng.random((40, 60)).astype(np.float32)
    assert np.array_equal(odf.resize_linear(im, 40, 60), im)                 # same size: identity
    assert np.allclose(odf.resize_linear(np.full((33, 47), 3.5, np.float32), 112, 112), 3.5)
    ramp = np.tile(np.arange(64, dtype=np.float32), (8, 1))
    half = odf.resize_linear(ramp, 8, 32)                                    # 2x down: pixel centres fall between two columns
    assert np.allclose(half[0], np.arange(32) * 2 + 0.5)
    up = odf.resize_linear(ramp[:, :4], 8, 8)                                # 2x up: borders clamp
    assert np.allclose(up[0], [0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3])
    fr = rng.integers(0, 256, (30, 50, 3), dtype=np.uint8)
    out = odf.mapf_frame(fr, 16, 16)
    assert out.shape == (16, 16, 3) and out.dtype == np.float32
    assert out.min() >= (0 - 102) / 255.0 - 1e-6 and out.max() <= (255 - 90) / 255.0 + 1e-6


def _resize_u8_scalar(im, H, W):
    """cv::resize 8-bit INTER_LINEAR, one pixel at a time with Python integers -- written separately from the vectorised
    oracle (oracle/dataflow.py::resize_linear_u8) so that the two restatements check each other."""
    import math
    H0, W0 = im.shape
    out = np.zeros((H, W), np.uint8)
    def table(dst, src, clamp_weight):
        t = []
        for d in range(dst):
            f = np.float32((d + 0.5) * (src / dst) - 0.5)
            s = math.floor(f)
            f = np.float32(f - np.float32(s))
            if clamp_weight and s < 0: s, f = 0, np.float32(0)
            if clamp_weight and s >= src - 1: s, f = src - 1, np.float32(0)
            w0 = int(np.rint(np.float32(np.float32(1) - f) * np.float32(2048)))
            w1 = int(np.rint(f * np.float32(2048)))
            t.append((min(max(s, 0), src - 1), min(max(s + 1, 0), src - 1), w0, w1))
        return t
    tx, ty = table(W, W0, True), table(H, H0, False)
    for y, (y0, y1, b0, b1) in enumerate(ty):
        for x, (x0, x1, a0, a1) in enumerate(tx):
            r0 = int(im[y0, x0]) * a0 + int(im[y0, x1]) * a1
            r1 = int(im[y1, x0]) * a0 + int(im[y1, x1]) * a1
            out[y, x] = min(max((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2, 0), 255)
    return out


def test_oracle_density_resize_is_opencvs_fixed_point_path():
    """dataflow.py:210-214 resizes the uint8 density map before dividing by 255: the targets are k / 255.
    Hand-computed vectors (11-bit weights, (>> 4, >> 16, + 2, >> 2) vertical pass):
      [0, 255] -> width 4: weights (2048,0) (1536,512) (512,1536) (2048,0); rows 0, 130560, 391680, 522240 -> 0, 64, 191, 255
      [10, 20, 30, 40] -> width 2 (2x down, weight 1024 each): (10+20)*1024 = 30720 -> 15, (30+40)*1024 -> 35
      column [0, 100] -> height 3: f = -1/6 (rows clip, weights 1707 / 341 on the same row 0) -> 0;
                                     f = 0.5 -> (1024*(0>>4)>>16) + (1024*(204800>>4)>>16) = 200 -> (200+2)>>2 = 50;
                                     f = 1 + 1/6 -> floor 1, f' = 1/6: rows (1, clip 2 -> 1), weights 1707 + 341 -> 100"""
    got = odf.resize_linear_u8(np.array([[0, 255]], np.uint8), 1, 4)
    assert got.tolist() == [[0, 64, 191, 255]]
    assert odf.resize_linear_u8(np.array([[10, 20, 30, 40]], np.uint8), 1, 2).tolist() == [[15, 35]]
    assert odf.resize_linear_u8(np.array([[0], [100]], np.uint8), 3, 1).tolist() == [[0], [50], [100]]
    rng = np.random.default_rng(3)
    for (H0, W0, H, W) in ((9, 13, 5, 7), (5, 7, 9, 13), (30, 50, 16, 16), (16, 16, 16, 16), (7, 31, 12, 4)):
        im = rng.integers(0, 256, (H0, W0), dtype=np.uint8)
        assert np.array_equal(odf.resize_linear_u8(im, H, W), _resize_u8_scalar(im, H, W)), (H0, W0, H, W)
    flat = odf.resize_linear_u8(np.full((33, 47), 200, np.uint8), 112, 112)
    assert flat.min() == 200 and flat.max() == 200
    d = odf.mapf_density(rng.integers(0, 256, (40, 60), dtype=np.uint8), 16, 16)
    assert d.dtype == np.float32 and np.array_equal(d, (np.round(d.astype(np.float64) * 255) / 255.0).astype(np.float32))     # k / 255 exactly


# ---- HIP kernels against the oracle --------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_cc_sim_nss_match_oracle():
    from sap3d_tensorflow_amd import metrics as gm
    s, t, f = _maps(10, n=8)
    cc = gm.CC_batch(s, t); sim = gm.SIM_batch(s, t); nss = gm.NSS_batch(s, f)
    for i in range(8):
        assert cc[i] == pytest.approx(om.CC(s[i], t[i]), rel=1e-9, abs=1e-12)
        assert sim[i] == pytest.approx(om.SIM(s[i], t[i]), rel=1e-10)
        assert nss[i] == pytest.approx(om.NSS(s[i], f[i]), rel=1e-9, abs=1e-12)
    assert gm.CC(s[0], s[0]) == pytest.approx(1.0, abs=1e-12)
    assert gm.SIM(s[0], t[0]) == pytest.approx(om.SIM(s[0], t[0]), rel=1e-10)
    # degenerate maps behave like the reference: flat map -> NaN, nothing fixated -> NaN
    assert np.isnan(gm.CC(np.ones((112, 112), np.float32), s[0]))
    assert np.isnan(gm.SIM(np.ones((112, 112), np.float32), s[0])) and np.isnan(om.SIM(np.ones((112, 112), np.float32), s[0]))
    assert np.isnan(gm.NSS(s[0], np.zeros((112, 112), np.float32)))
    with pytest.raises(ValueError):
        gm.CC(s[0], t[0][:56])


@pytest.mark.gpu
def test_auc_judd_matches_oracle():
    from sap3d_tensorflow_amd import metrics as gm
    s, _, f = _maps(11, n=6)
    s[1] = np.floor(s[1] * 16) / 16                     # ties among thresholds and pixels
    f[2] = 0                                            # no fixation -> NaN
    f[3] = 0; f[3, 40, 41] = 1                          # a single fixation
    f[4] = (s[4] > 0.5)                                 # half the map fixated, like a density map > 0.5 (train.py:260)
    got = gm.AUC_Judd_batch(s, f, jitter=False)
    for i in range(6):
        want = om.AUC_Judd(s[i], f[i])
        if np.isnan(want):
            assert np.isnan(got[i])
        else:
            assert got[i] == pytest.approx(want, abs=1e-12)
    # the reference's jitter, supplied explicitly
    jit = (np.random.default_rng(5).random(s.shape) * 1e-7).astype(np.float32)
    got = gm.AUC_Judd_batch(s, f, jitter=jit)
    for i in (0, 1, 4):
        assert got[i] == pytest.approx(om.AUC_Judd(s[i], f[i], jitter=jit[i]), abs=1e-12)
    # a map at the evaluation resolution of test.py (1080 x 960)
    rng = np.random.default_rng(6)
    big = rng.random((1080, 960)).astype(np.float32)
    bf = (rng.random((1080, 960)) < 0.002).astype(np.float32)
    assert gm.AUC_Judd(big, bf, jitter=False) == pytest.approx(om.AUC_Judd(big, bf), abs=1e-12)
    # every pixel fixated: the false-positive rate divides by zero in the reference too (NaN), no crash
    allfix = np.ones((16, 16), np.float32)
    small = rng.random((16, 16)).astype(np.float32)
    with np.errstate(all="ignore"):
        assert np.isnan(om.AUC_Judd(small, allfix))
    assert np.isnan(gm.AUC_Judd(small, allfix, jitter=False))
    # a 1 x N map and a single-row batch
    line = rng.random((1, 300)).astype(np.float32); lf = (rng.random((1, 300)) < 0.1).astype(np.float32)
    assert gm.AUC_Judd(line, lf, jitter=False) == pytest.approx(om.AUC_Judd(line, lf), abs=1e-12)
    # two runs are bit-identical (integer counters, fixed-order sums)
    assert gm.AUC_Judd(big, bf, jitter=False) == gm.AUC_Judd(big, bf, jitter=False)


@pytest.mark.gpu
def test_auc_borji_matches_oracle():
    from sap3d_tensorflow_amd import metrics as gm
    s, _, f = _maps(12, n=1)
    s, f = s[0], f[0]
    n_fix = int(f.sum())
    r = np.random.default_rng(7).integers(0, s.size, (n_fix, 20))
    want, per = om.AUC_Borji(s, f, r, 0.1)
    assert gm.AUC_Borji(s, f, n_rep=20, step_size=0.1, rand_idx=r) == pytest.approx(want, abs=1e-12)
    want2, _ = om.AUC_Borji(s, f, r, 0.03)
    assert gm.AUC_Borji(s, f, n_rep=20, step_size=0.03, rand_idx=r) == pytest.approx(want2, abs=1e-12)
    assert np.isnan(gm.AUC_Borji(s, np.zeros_like(f)))
    np.random.seed(3)                                   # default: numpy's global generator, as the reference draws
    a = gm.AUC_Borji(s, f, n_rep=10)
    np.random.seed(3)
    rr = np.random.randint(0, s.size, [n_fix, 10])
    assert a == pytest.approx(om.AUC_Borji(s, f, rr, 0.1)[0], abs=1e-12)


@pytest.mark.gpu
def test_mapf_matches_oracle():
    from sap3d_tensorflow_amd import dataflow as gdf
    rng = np.random.default_rng(20)
    for (H0, W0) in ((270, 480), (56, 56), (112, 112), (225, 401)):
        frames = rng.integers(0, 256, (3, H0, W0, 3), dtype=np.uint8)
        got = gdf.mapf_frames(frames, 112)
        want = np.stack([odf.mapf_frame(fr, 112, 112) for fr in frames])
        assert got.shape == (3, 112, 112, 3)
        assert np.abs(got - want).max() <= 1e-6, (H0, W0, np.abs(got - want).max())
        dens = rng.integers(0, 256, (2, H0, W0), dtype=np.uint8)
        gd = gdf.mapf_density(dens, 112)
        wd = np.stack([odf.mapf_density(d, 112, 112) for d in dens])
        assert np.array_equal(gd, wd), (H0, W0)                 # byte arithmetic (uint8 fixed-point resize): bit-exact
    for (H0, W0, H, W) in ((5, 7, 9, 13), (9, 13, 5, 7), (1, 2, 1, 4), (2, 1, 3, 1)):          # upscaling: clamped borders, clipped rows
        dens = rng.integers(0, 256, (2, H0, W0), dtype=np.uint8)
        assert np.array_equal(gdf.mapf_density(dens, (H, W)), np.stack([odf.mapf_density(d, H, W) for d in dens])), (H0, W0, H, W)
    with pytest.raises(ValueError):
        gdf.mapf_frames(np.zeros((2, 8, 8), np.uint8))
