"""Label Distribution Smoothing weights (host) — the arithmetic of ``datasets.py:55-83``.

``prepare_weights`` is what ``IMDBWIKI._prepare_weights`` / ``AgeDB._prepare_weights`` call; the numbers come
from the native host routine ``dir_lds_weights`` in ``libdir_hip.so`` (bit-exact with numpy/scipy's
accumulation orders); the smoothing window comes from scipy on the host like the reference.
"""
import logging

import numpy as np

from . import ops
from .utils import get_lds_kernel_window

print = logging.info


def prepare_weights(labels, reweight, max_target=121, lds=False, lds_kernel='gaussian', lds_ks=5, lds_sigma=2):
    """Returns a list of ``np.float32`` (like the reference) or ``None`` when ``reweight == 'none'``."""
    assert reweight in {'none', 'inverse', 'sqrt_inv'}
    assert reweight != 'none' if lds else True, \
        "Set reweight to \'sqrt_inv\' (default) or \'inverse\' when using LDS"
    labels = np.asarray(labels)
    if not len(labels) or reweight == 'none':
        return None
    print(f"Using re-weighting: [{reweight.upper()}]")
    window = None
    if lds:
        window = get_lds_kernel_window(lds_kernel, lds_ks, lds_sigma)
        print(f'Using LDS: [{lds_kernel.upper()}] ({lds_ks}/{lds_sigma})')
    return list(ops.lds_weights(labels, max_target, reweight, lds, window))
