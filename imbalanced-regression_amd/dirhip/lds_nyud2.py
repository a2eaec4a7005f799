"""Per-pixel LDS loss weights of NYUD2-DIR — ``nyud2-dir/loaddata.py:11-19,29-67`` (``depthDataset._get_bucket_weights``,
``get_bin_idx``, ``_get_weights``). SURVEY.md §8f-1.

The reference keeps the training set's depth histogram as a constant (``TRAIN_BUCKET_NUM``: pixels per 0.1 m bucket, 100
buckets, the first 7 empty), smooths it with the LDS window (``convolve1d(mode='reflect')`` over the buckets from
``bucket_start`` on), and maps every pixel's depth to ``scaling / smoothed[bucket]``. Host part (100 numbers, once per run):
the same numpy / scipy calls as the reference, so it is bit-exact by construction (the windows too). Device
part (one weight per pixel, every batch): bucket = min(int(depth * 10), 99) by the hand-written ``dir_fds_bin_scaled``
kernel + a table lookup, instead of the reference's Python ``map`` over 17 328 pixels per image on the host.
"""
import logging

import numpy as np
import torch
from scipy.ndimage import convolve1d

from . import ops
from .utils import get_lds_kernel_window

print = logging.info

# nyud2-dir/loaddata.py:11-19 — pixels of the NYUD2-DIR training set per 0.1 m depth bucket (a dataset statistic the
# reference ships as a constant "for data loading efficiency"; buckets 0-6 are empty: no depth below 0.7 m)
TRAIN_BUCKET_NUM = [0, 0, 0, 0, 0, 0, 0, 25848691, 24732940, 53324326, 69112955, 54455432, 95637682, 71403954, 117244217,
                    84813007, 126524456, 84486706, 133130272, 95464874, 146051415, 146133612, 96561379, 138366677, 89680276,
                    127689043, 81608990, 119121178, 74360607, 106839384, 97595765, 66718296, 90661239, 53103021, 83340912,
                    51365604, 71262770, 42243737, 65860580, 38415940, 53647559, 54038467, 28335524, 41485143, 32106001,
                    35936734, 23966211, 32018765, 19297203, 31503743, 21681574, 16363187, 25743420, 12769509, 17675327,
                    13147819, 15798560, 9547180, 14933200, 9663019, 12887283, 11803562, 7656609, 11515700, 7756306, 9046228,
                    5114894, 8653419, 6859433, 8001904, 6430700, 3305839, 6318461, 3486268, 5621065, 4030498, 3839488, 3220208,
                    4483027, 2555777, 4685983, 3145082, 2951048, 2762369, 2367581, 2546089, 2343867, 2481579, 1722140, 3018892,
                    2325197, 1952354, 2047038, 1858707, 2052729, 1348558, 2487278, 1314198, 3338550, 1132666]


def get_bucket_weights(reweight, lds=False, lds_kernel='gaussian', lds_ks=5, lds_sigma=2, bucket_num=100, bucket_start=7):
    """loaddata.py:29-55: list of ``bucket_num`` ``np.float32`` bucket weights, or ``None`` for ``reweight == 'none'``."""
    assert reweight in {'none', 'inverse', 'sqrt_inv'}
    assert reweight != 'none' if lds else True, "Set reweight to \'sqrt_inv\' or \'inverse\' (default) when using LDS"
    if reweight == 'none':
        return None
    print(f"Using re-weighting: [{reweight.upper()}]")
    if lds:
        value_lst = TRAIN_BUCKET_NUM[bucket_start:]
        window = get_lds_kernel_window(lds_kernel, lds_ks, lds_sigma)
        print(f'Using LDS: [{lds_kernel.upper()}] ({lds_ks}/{lds_sigma})')
        if reweight == 'sqrt_inv':
            value_lst = np.sqrt(value_lst)
        smoothed = convolve1d(np.asarray(value_lst), weights=window, mode='reflect')      # integer-preserving for 'inverse' (A.5)
        smoothed = [smoothed[0]] * bucket_start + list(smoothed)
        scaling = np.sum(TRAIN_BUCKET_NUM) / np.sum(np.array(TRAIN_BUCKET_NUM) / np.array(smoothed))
        return [np.float32(scaling / smoothed[b]) for b in range(bucket_num)]
    value_lst = [TRAIN_BUCKET_NUM[bucket_start]] * bucket_start + TRAIN_BUCKET_NUM[bucket_start:]
    if reweight == 'sqrt_inv':
        value_lst = np.sqrt(value_lst)
    scaling = np.sum(TRAIN_BUCKET_NUM) / np.sum(np.array(TRAIN_BUCKET_NUM) / np.array(value_lst))
    return [np.float32(scaling / value_lst[b]) for b in range(bucket_num)]


def get_bin_idx(x):
    """loaddata.py:57-58."""
    return min(int(x * np.float32(10)), 99)


class PixelWeights:
    """``depthDataset._get_weights`` (loaddata.py:60-69) for device depth maps: ``weights(depth)`` returns a float32 tensor
    of ``depth``'s shape with ``bucket_weights[min(int(d * 10), 99)]`` per pixel (all ones when there is no re-weighting)."""

    def __init__(self, bucket_weights):
        self.bucket_weights = bucket_weights
        self._table = None

    def weights(self, depth):
        if self.bucket_weights is None:
            return torch.ones_like(depth, dtype=torch.float32)
        if depth.dtype != torch.float32:
            raise TypeError("depth must be float32 (the reference asserts it, loaddata.py:64)")
        if self._table is None or self._table.device != depth.device:
            self._table = torch.tensor(np.asarray(self.bucket_weights, dtype=np.float32), device=depth.device)
        flat = depth.contiguous().view(-1)
        bins = ops.bin_scaled(flat, 10.0, 0, len(self.bucket_weights))       # clamp(int(d * 10), 0, 99), float32 product
        return self._table[bins.long()].view(depth.shape)
