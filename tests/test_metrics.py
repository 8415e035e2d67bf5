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
        assert np.abs(gd - wd).max() <= 1e-6
    with pytest.raises(ValueError):
        gdf.mapf_frames(np.zeros((2, 8, 8), np.uint8))
