"""ORACLE (test infrastructure only -- never imported by the product): numpy restatement of the reference's saliency
metrics, /root/reference/utils/metrics.py:25-287 with the `normalize` helper of /root/reference/utils/metric_utils.py:10-53.

PARITY UNPINNED: the reference module cannot be imported here (its top-level `import cv2` and `from skimage import ...`
have no module in this image; nothing was refused, the packages are simply absent) and the reference ships no metric
fixtures, so nothing but a side-by-side reading pins this file.  Differences from the reference, all deliberate:
  * inputs are cast to float64 first (the reference works in whatever dtype it is given; test.py:160-183 feeds float64
    density maps and float32 predictions);
  * maps must have one shape (the reference's skimage `resize` branch is not restated);
  * the random draws are arguments: AUC_Judd's jitter (`random.rand(*shape) * 1e-7`, :62-63) and AUC_Borji's pixel
    indices (`random.randint(0, n_pixels, [n_fix, n_rep])`, :139).
"""
import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz        # np.trapz(y, x), utils/metrics.py:85,153


def normalize(x, method="standard"):
    """utils/metric_utils.py:41-49 (axis=None branch)."""
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        if method == "standard":
            return (x - np.mean(x)) / np.std(x)
        if method == "range":
            return (x - np.min(x)) / (np.max(x) - np.min(x))
        if method == "sum":
            return x / float(np.sum(x))
    raise ValueError('method not in {"standard", "range", "sum"}')


def _same_shape(a, b):
    if a.shape != b.shape:
        raise ValueError("maps of different shape: the reference's resize branch is not restated")


def AUC_Judd(saliency_map, fixation_map, jitter=None):
    """utils/metrics.py:25-85.  jitter: None (jitter=False) or the noise array the reference would add (:62-63)."""
    S2 = np.array(saliency_map, dtype=np.float32)
    F2 = np.asarray(fixation_map) > 0.5
    _same_shape(S2, F2)
    if not np.any(F2):
        return np.nan                                   # :58-60
    if jitter is not None:
        S2 = S2 + np.asarray(jitter, dtype=np.float32)    # :62-63 (float32 sum, as the product adds it)
    S = S2.ravel().astype(np.float64)
    F = F2.ravel()
    S_fix = S[F]
    n_fix = len(S_fix)
    n_pixels = len(S)
    thresholds = sorted(S_fix, reverse=True)            # :76
    tp = np.zeros(len(thresholds) + 2)
    fp = np.zeros(len(thresholds) + 2)
    tp[0] = 0; tp[-1] = 1
    fp[0] = 0; fp[-1] = 1
    S_sorted = np.sort(S)
    for k, thresh in enumerate(thresholds):
        above_th = n_pixels - np.searchsorted(S_sorted, thresh, side="left")      # == np.sum(S >= thresh), :82
        tp[k + 1] = (k + 1) / float(n_fix)
        fp[k + 1] = (above_th - k - 1) / float(n_pixels - n_fix)
    return _trapz(tp, fp)


def AUC_Borji(saliency_map, fixation_map, rand_idx, step_size=0.1):
    """utils/metrics.py:88-154 with r = rand_idx [n_fix, n_rep] (:139).  Returns (mean AUC, per-split AUCs)."""
    S2 = np.asarray(saliency_map, dtype=np.float64)
    F2 = np.asarray(fixation_map) > 0.5
    _same_shape(S2, F2)
    if not np.any(F2):
        return np.nan, None
    S = normalize(S2, method="range").ravel()           # :129
    F = F2.ravel()
    S_fix = S[F]
    n_fix = len(S_fix)
    r = np.asarray(rand_idx)
    n_rep = r.shape[1]
    if r.shape[0] != n_fix:
        raise ValueError("rand_idx must be [n_fix, n_rep]")
    S_rand = S[r]
    trapz = _trapz
    auc = np.zeros(n_rep) * np.nan
    for rep in range(n_rep):
        thresholds = np.r_[0:np.max(np.r_[S_fix, S_rand[:, rep]]):step_size][::-1]
        tp = np.zeros(len(thresholds) + 2)
        fp = np.zeros(len(thresholds) + 2)
        tp[0] = 0; tp[-1] = 1
        fp[0] = 0; fp[-1] = 1
        for k, thresh in enumerate(thresholds):
            tp[k + 1] = np.sum(S_fix >= thresh) / float(n_fix)
            fp[k + 1] = np.sum(S_rand[:, rep] >= thresh) / float(n_fix)
        auc[rep] = trapz(tp, fp)
    return np.mean(auc), auc


def NSS(saliency_map, fixation_map):
    """utils/metrics.py:200-224."""
    s_map = np.asarray(saliency_map, dtype=np.float64)
    f_map = np.asarray(fixation_map) > 0.5
    _same_shape(s_map, f_map)
    s_map = normalize(s_map, method="standard")
    with np.errstate(invalid="ignore"):
        return np.mean(s_map[f_map]) if np.any(f_map) else np.nan


def CC(saliency_map1, saliency_map2):
    """utils/metrics.py:227-250."""
    map1 = np.asarray(saliency_map1, dtype=np.float64)
    map2 = np.asarray(saliency_map2, dtype=np.float64)
    _same_shape(map1, map2)
    map1 = normalize(map1, method="standard")
    map2 = normalize(map2, method="standard")
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.corrcoef(map1.ravel(), map2.ravel())[0, 1]


def SIM(saliency_map1, saliency_map2):
    """utils/metrics.py:258-287."""
    map1 = np.asarray(saliency_map1, dtype=np.float64)
    map2 = np.asarray(saliency_map2, dtype=np.float64)
    _same_shape(map1, map2)
    map1 = normalize(normalize(map1, method="range"), method="sum")
    map2 = normalize(normalize(map2, method="range"), method="sum")
    return np.sum(np.minimum(map1, map2))
