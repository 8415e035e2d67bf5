"""The reference's validation metrics (utils/metrics.py) on the GPU: same names, same argument meaning, computed by
libp3dhip (csrc/metrics.hip) in float64 on float32 maps.  `*_batch` variants take [n_maps, H, W] stacks -- one launch
for all the last-frame maps of a validation pass (train.py:249-260).  Maps must share one shape: the reference's
resize-to-match branch (skimage) is not reproduced and raises here."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def _maps(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        raise ValueError("maps of different shape %s / %s: resize them first (the reference's skimage branch is not "
                         "part of this library)" % (a.shape, b.shape))
    if a.ndim < 2 or a.size == 0:
        raise ValueError("expected non-empty [H, W] or [n, H, W] maps")
    return a, b


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _two_map(fn, m1, m2, batched, device):
    a, b = _maps(m1, m2)
    n = a.shape[0] if batched else 1
    out = np.empty(n, np.float64)
    check(fn(device, _fp(a), _fp(b), n, a.size // n, _dp(out)))
    return out if batched else float(out[0])


def CC(saliency_map1, saliency_map2, device=0):
    """Pearson correlation of two maps (utils/metrics.py:227-250)."""
    return _two_map(lib().p3d_metric_cc, saliency_map1, saliency_map2, False, device)


def SIM(saliency_map1, saliency_map2, device=0):
    """Histogram intersection of two maps scaled to [0,1] and to sum 1 (utils/metrics.py:258-287)."""
    return _two_map(lib().p3d_metric_sim, saliency_map1, saliency_map2, False, device)


def NSS(saliency_map, fixation_map, device=0):
    """Mean standardised saliency at fixated pixels, fixation_map > 0.5 (utils/metrics.py:200-224)."""
    return _two_map(lib().p3d_metric_nss, saliency_map, np.asarray(fixation_map, dtype=np.float32), False, device)


def CC_batch(maps1, maps2, device=0):
    return _two_map(lib().p3d_metric_cc, maps1, maps2, True, device)


def SIM_batch(maps1, maps2, device=0):
    return _two_map(lib().p3d_metric_sim, maps1, maps2, True, device)


def NSS_batch(saliency_maps, fixation_maps, device=0):
    return _two_map(lib().p3d_metric_nss, saliency_maps, np.asarray(fixation_maps, dtype=np.float32), True, device)


def _judd(sal, fix, jitter, batched, device, rng):
    a, b = _maps(sal, np.asarray(fix, dtype=np.float32))
    n = a.shape[0] if batched else 1
    if jitter is True:          # the reference's default: saliency_map += random.rand(*shape) * 1e-7 (utils/metrics.py:62-63)
        jitter = (rng if rng is not None else np.random).random_sample(a.shape) * 1e-7
    jit = None
    if jitter is not None and jitter is not False:
        jit = np.ascontiguousarray(jitter, dtype=np.float32)
        if jit.shape != a.shape:
            raise ValueError("jitter must have the maps' shape")
    out = np.empty(n, np.float64)
    check(lib().p3d_metric_auc_judd(device, _fp(a), _fp(b), _fp(jit) if jit is not None else None, n, a.size // n, _dp(out)))
    return out if batched else float(out[0])


def AUC_Judd(saliency_map, fixation_map, jitter=True, device=0, rng=None):
    """Area under the ROC curve swept over the saliency values at fixated pixels (utils/metrics.py:25-85).  jitter: True
    (draw the reference's 1e-7 noise from numpy), False, or the noise array itself.  NaN when nothing is fixated."""
    return _judd(saliency_map, fixation_map, jitter, False, device, rng)


def AUC_Judd_batch(saliency_maps, fixation_maps, jitter=True, device=0, rng=None):
    return _judd(saliency_maps, fixation_maps, jitter, True, device, rng)


def AUC_Borji(saliency_map, fixation_map, n_rep=100, step_size=0.1, rand_sampler=None, device=0, rng=None, rand_idx=None):
    """utils/metrics.py:88-154.  The random pixel indices come from numpy exactly as in the reference
    (random.randint(0, n_pixels, [n_fix, n_rep]), :139) unless `rand_idx` supplies them; `rand_sampler` (the hook
    AUC_shuffled uses, :145) is not supported."""
    if rand_sampler is not None:
        raise NotImplementedError("rand_sampler (AUC_shuffled) is outside the ported path")
    a, b = _maps(saliency_map, np.asarray(fixation_map, dtype=np.float32))
    n_fix = int(np.count_nonzero(b > 0.5))
    if n_fix == 0:
        return float("nan")                 # 'no fixation to predict' (utils/metrics.py:122-124)
    if rand_idx is None:
        rand_idx = (rng if rng is not None else np.random).randint(0, a.size, [n_fix, n_rep])
    r = np.ascontiguousarray(rand_idx, dtype=np.int32)
    if r.shape != (n_fix, n_rep):
        raise ValueError("rand_idx must be [n_fix, n_rep] = [%d, %d]" % (n_fix, n_rep))
    out = np.empty(n_rep, np.float64)
    check(lib().p3d_metric_auc_borji(device, _fp(a), _fp(b), r.ctypes.data_as(C.POINTER(C.c_int)), a.size, n_fix, n_rep,
                                     float(step_size), _dp(out)))
    return float(np.mean(out))
