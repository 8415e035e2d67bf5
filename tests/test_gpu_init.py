"""the initialiser law of `get_conv_weight` (p3d.py:10-16: xavier_initializer, also for the
convS / convT biases), of tf.layers (zeros biases), of BatchNorm / GroupNorm (gamma 1, beta 0, moving mean 0, variance 1)
and of CBAM's dense / conv layers (variance_scaling_initializer, utils/network.py:212-213,264: truncated normal).  The
random STREAM of TensorFlow cannot be matched; the LAW can: support, mean and variance of every variable."""
import numpy as np
import pytest

from oracle import p3d

pytestmark = pytest.mark.gpu


def _fans(shape):
    if len(shape) == 1:
        return shape[0], shape[0]
    rf = int(np.prod(shape[:-2]))
    return rf * shape[-2], rf * shape[-1]


def test_xavier_and_constant_laws_unet():
    from sap3d_tensorflow_amd import P3DSession
    s = P3DSession("unet", batch=1, frames=16, height=32, width=32, seed=7)          # reference architecture (base 64, 3/8/36)
    seen = 0
    for name, shape, trainable in s.variables():
        v = s.get_param(name).astype(np.float64)
        if name.endswith("/gamma") or name.endswith("/moving_variance"):
            assert np.all(v == 1.0), name
        elif name.endswith("/beta") or name.endswith("/moving_mean") or (name.endswith("/bias") and "_S_" not in name and "_T_" not in name):
            assert np.all(v == 0.0), name                                            # tf.layers biases: zeros
        else:                                                                         # get_conv_weight: kernels AND the ST biases
            fi, fo = _fans(shape)
            L = np.sqrt(6.0 / (fi + fo))
            assert np.abs(v).max() <= L * (1 + 1e-6), name
            if v.size >= 4096:
                assert abs(v.mean()) <= 4 * L / np.sqrt(3 * v.size), name            # mean 0 within 4 sigma
                assert v.var() == pytest.approx(L * L / 3, rel=0.08), name           # variance of U(-L, L)
                assert np.abs(v).max() >= 0.98 * L, name                              # the support is used up to its edge
                seen += 1
    assert seen > 150
    a = s.get_param("conv3_20_1")
    s.init_params(7)
    assert np.array_equal(a, s.get_param("conv3_20_1"))                              # same seed, same draw
    s.init_params(8)
    assert not np.array_equal(a, s.get_param("conv3_20_1"))
    s.close()


def test_cbam_variance_scaling_is_a_truncated_normal():
    from sap3d_tensorflow_amd import P3DSession
    s = P3DSession("gn_p3d", batch=1, frames=16, height=32, width=32, seed=3)
    checked = 0
    for name, shape, _ in s.variables():
        if "cbam_" not in name or not name.endswith("kernel"):
            continue
        v = s.get_param(name).astype(np.float64)
        fan_in = _fans(shape)[0]
        sd = np.sqrt(1.3 * 2.0 / fan_in)                                             # stddev handed to tf.truncated_normal
        assert np.abs(v).max() <= 2 * sd * (1 + 1e-6), name                          # truncated at two standard deviations
        if v.size >= 8192:
            assert abs(v.mean()) <= 5 * sd / np.sqrt(v.size), name
            assert v.std() == pytest.approx(0.87962566 * sd, rel=0.05), name         # std of N(0,1) truncated to [-2, 2]
            kurt = ((v - v.mean()) ** 4).mean() / v.var() ** 2
            assert 2.0 < kurt < 2.6, (name, kurt)                                    # a uniform would give 1.8, a normal 3.0
            checked += 1
    assert checked >= 10
    s.close()
