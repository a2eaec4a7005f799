"""TEST INFRASTRUCTURE ONLY — restatement of the reference LDS weight preparation.

Follows ``imdb-wiki-dir/utils.py:110-122`` (``get_lds_kernel_window``) and
``imdb-wiki-dir/datasets.py:55-83`` (``_prepare_weights``; identical in ``agedb-dir``).
The arithmetic that decides the bits lives in third-party libraries the reference does
not pin (``README.md:21-23``): scipy.ndimage ``gaussian_filter1d`` / ``convolve1d`` and
numpy scalar/array semantics. De-facto pin = the versions in this image (scipy 1.15.3,
numpy 2.2.6); this oracle calls the same library routines with the same dtypes, vectorised,
so it reproduces the reference bit for bit. Pinned against golden vectors (real AgeDB
label histogram + synthetic label sets) in ``tests/test_oracle_golden.py``.
"""
import numpy as np
from scipy.ndimage import convolve1d, gaussian_filter1d
from scipy.signal.windows import triang


def get_lds_kernel_window(kernel, ks, sigma):
    """utils.py:110-122: float64, gaussian/laplace normalised by MAX, triang un-normalised."""
    assert kernel in ("gaussian", "triang", "laplace")
    half = (ks - 1) // 2
    if kernel == "gaussian":
        delta = [0.0] * half + [1.0] + [0.0] * half            # python-list delta -> float64
        g = gaussian_filter1d(delta, sigma=sigma)
        return g / max(g)                                       # utils.py:115
    if kernel == "triang":
        return triang(ks)                                       # utils.py:117
    lap = [np.exp(-abs(x) / sigma) / (2.0 * sigma) for x in np.arange(-half, half + 1)]
    return np.asarray(lap) / max(lap)                           # utils.py:119-120


def label_bins(labels, max_target=121):
    """datasets.py:63,68,78: ``min(max_target - 1, int(label))`` (truncation toward zero)."""
    lab = np.asarray(labels)
    return np.minimum(max_target - 1, np.trunc(lab).astype(np.int64))


def prepare_weights(labels, reweight, max_target=121, lds=False, lds_kernel="gaussian",
                    lds_ks=5, lds_sigma=2):
    """datasets.py:55-83. Returns float32 array [N] or None."""
    assert reweight in {"none", "inverse", "sqrt_inv"}
    assert reweight != "none" if lds else True
    bins = label_bins(labels, max_target)
    counts = np.bincount(bins, minlength=max_target)[:max_target].astype(np.int64)   # :60-63
    if reweight == "sqrt_inv":
        value = np.sqrt(counts)                                 # :64-65 float64
    elif reweight == "inverse":
        value = np.clip(counts, 5, 1000)                        # :66-67 stays int64 (A.5)
    else:
        value = counts
    if len(bins) == 0 or reweight == "none":                    # :69-70
        return None
    if lds:                                                     # :73-78
        window = get_lds_kernel_window(lds_kernel, lds_ks, lds_sigma)
        value = convolve1d(value, weights=window, mode="constant")   # dtype-preserving (A.5)
    num_per_label = value[bins]
    with np.errstate(divide="ignore"):
        w = (1.0 / num_per_label.astype(np.float64)).astype(np.float32)   # :80 np.float32(1 / x)
    total = np.sum(w)                                           # :81 float32 pairwise (A.6 / E.2)
    scaling = len(w) / total                                    # python int / np.float32 -> float32
    return (scaling * w).astype(np.float32)                     # :82
