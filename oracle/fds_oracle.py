"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference FDS (age variant).

Follows ``imdb-wiki-dir/fds.py`` (= ``agedb-dir/fds.py`` except the default
``bucket_start``, line 16) and ``utils.py:97-107`` of the upstream reference.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``imbalanced-regression_amd/``)
never does.

Parity status: PINNED — ``tests/test_oracle_golden.py`` checks every function
below against golden vectors produced by running the reference's own code in
the build container (``tests/golden/gen_golden.py``), and against the live
reference when ``/root/reference`` is present.

Arithmetic: float32 wherever the reference computes in float32 (momentum blend,
smoothing, calibration — these are bit-faithful restatements op by op), except
the per-bin mean/variance, which the reference obtains from torch's float32
cascade/Welford reductions and the oracle computes in float64 and rounds to
float32 (the pinned difference to the reference is <= 2e-6 relative).
"""
import numpy as np
from scipy.ndimage import gaussian_filter1d
from scipy.signal.windows import triang

F32 = np.float32


def fds_kernel_window(kernel, ks, sigma):
    """Reference ``FDS._get_kernel_window`` (fds.py:37-52): float32, sum-normalised."""
    assert kernel in ("gaussian", "triang", "laplace")
    half = (ks - 1) // 2
    if kernel == "gaussian":
        delta = np.zeros(ks, dtype=np.float32)              # fds.py:42-43 (float32 delta)
        delta[half] = 1.0
        g = gaussian_filter1d(delta, sigma=sigma)
        win = g / sum(g)                                      # fds.py:44 (python sum)
    elif kernel == "triang":
        t = triang(ks)
        win = t / sum(t)                                      # fds.py:46
    else:
        lap = [np.exp(-abs(x) / sigma) / (2.0 * sigma) for x in np.arange(-half, half + 1)]
        win = np.asarray(lap) / sum(lap)                      # fds.py:48-49
    return np.asarray(win, dtype=np.float32)                  # fds.py:52 (torch.float32)


def calibrate_mean_var(matrix, m1, v1, m2, v2, clip_min=0.1, clip_max=10):
    """Reference ``utils.calibrate_mean_var`` (utils.py:97-107), float32, returns a new array
    (or the input object itself on the identity branch, like the reference)."""
    if np.sum(v1, dtype=np.float32) < 1e-10:                  # utils.py:98-99
        return matrix
    lo, hi = F32(clip_min), F32(clip_max)
    if (v1 == 0.0).any():                                     # utils.py:100-104
        valid = v1 != 0.0
        with np.errstate(all="ignore"):
            factor = np.clip(v2[valid] / v1[valid], lo, hi)
        out = matrix.copy()
        out[:, valid] = (matrix[:, valid] - m1[valid]) * np.sqrt(factor) + m2[valid]
        return out
    with np.errstate(all="ignore"):
        factor = np.clip(v2 / v1, lo, hi)                     # utils.py:106
    return (matrix - m1) * np.sqrt(factor) + m2               # utils.py:107


def calibrate_scale(v1, v2, clip_min=0.1, clip_max=10):
    """Per-column multiplier the reference applies (``sqrt(clamp(v2/v1))``); -1 marks the
    columns/rows it leaves untouched (v1 == 0, or sum(v1) < 1e-10). Same quantity the
    product's ``dir_fds_prepare_scale`` emits. v1, v2: [C]."""
    if np.sum(v1, dtype=np.float32) < 1e-10:
        return np.full(v1.shape, -1.0, dtype=np.float32)
    with np.errstate(all="ignore"):
        s = np.sqrt(np.clip(v2 / v1, F32(clip_min), F32(clip_max))).astype(np.float32)
    s[v1 == 0.0] = -1.0
    return s


def _row_groups(labels, bucket_start, bucket_num):
    """The reference's per-unique-label row selection (fds.py:91-99 and :120-137).
    Yields (label_value, boolean row mask) in sorted label order, skipping what the
    reference ``continue``s on. Boundary lumping depends on the boundary label being
    present in ``labels``."""
    for label in np.unique(labels):                            # sorted, like torch.unique
        if label > bucket_num - 1 or label < bucket_start:
            continue
        elif label == bucket_start:
            rows = labels <= label
        elif label == bucket_num - 1:
            rows = labels >= label
        else:
            rows = labels == label
        yield label, rows


def bin_index(labels, bucket_start, bucket_num):
    """Row -> table-row index (or -1 = row untouched) implied by ``_row_groups``; what the
    product's ``dir_fds_bin_index`` must produce (bit-exact, int32)."""
    labels = np.asarray(labels, dtype=np.float32).reshape(-1)
    bins = np.full(labels.shape, -1, dtype=np.int32)
    for label, rows in _row_groups(labels, bucket_start, bucket_num):
        bins[rows] = int(F32(label) - F32(bucket_start))       # fds.py:104 int(label - bucket_start)
    return bins


class FDSOracle:
    """State machine restating ``FDS`` (fds.py:14-144) with numpy buffers of the same names."""

    def __init__(self, feature_dim, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1,
                 kernel="gaussian", ks=5, sigma=2, momentum=0.9):
        self.feature_dim = feature_dim
        self.bucket_num = bucket_num
        self.bucket_start = bucket_start
        self.kernel_window = fds_kernel_window(kernel, ks, sigma)
        self.half_ks = (ks - 1) // 2
        self.momentum = momentum
        self.start_update = start_update
        self.start_smooth = start_smooth
        nb = bucket_num - bucket_start
        z = lambda: np.zeros((nb, feature_dim), dtype=np.float32)
        o = lambda: np.ones((nb, feature_dim), dtype=np.float32)
        self.epoch = np.full((1,), start_update, dtype=np.float32)          # fds.py:28
        self.running_mean, self.running_var = z(), o()                      # fds.py:29-30
        self.running_mean_last_epoch, self.running_var_last_epoch = z(), o()   # :31-32
        self.smoothed_mean_last_epoch, self.smoothed_var_last_epoch = z(), o()  # :33-34
        self.num_samples_tracked = np.zeros((nb,), dtype=np.float32)        # :35

    BUFFERS = ("epoch", "running_mean", "running_var", "running_mean_last_epoch",
               "running_var_last_epoch", "smoothed_mean_last_epoch",
               "smoothed_var_last_epoch", "num_samples_tracked")

    def state(self):
        return {k: np.array(getattr(self, k), copy=True) for k in self.BUFFERS}

    # ---- fds.py:54-67 -------------------------------------------------------------
    def smooth_bins(self, table):
        """ks-tap correlation along the bin axis with reflect ('mirror') padding, float32,
        taps accumulated left to right (== F.pad(reflect)+F.conv1d to <=2e-7)."""
        h = self.half_ks
        nb = table.shape[0]
        pad = np.pad(table, ((h, h), (0, 0)), mode="reflect") if h > 0 else table
        out = np.zeros_like(table)
        for k in range(2 * h + 1):
            out = out + self.kernel_window[k] * pad[k:k + nb]
        return out.astype(np.float32)

    def _update_last_epoch_stats(self):
        self.running_mean_last_epoch = self.running_mean       # fds.py:55-56: ALIAS (A.1)
        self.running_var_last_epoch = self.running_var
        self.smoothed_mean_last_epoch = self.smooth_bins(self.running_mean_last_epoch)
        self.smoothed_var_last_epoch = self.smooth_bins(self.running_var_last_epoch)

    def reset(self):                                           # fds.py:69-76
        self.running_mean[...] = 0
        self.running_var[...] = 1
        self.running_mean_last_epoch[...] = 0
        self.running_var_last_epoch[...] = 1
        self.smoothed_mean_last_epoch[...] = 0
        self.smoothed_var_last_epoch[...] = 1
        self.num_samples_tracked[...] = 0

    def update_last_epoch_stats(self, epoch):                  # fds.py:78-82
        if epoch == self.epoch[0] + 1:
            self.epoch += 1
            self._update_last_epoch_stats()

    def update_running_stats(self, features, labels, epoch):   # fds.py:84-113
        if epoch < self.epoch[0]:
            return
        features = np.asarray(features, dtype=np.float32)
        labels = np.asarray(labels, dtype=np.float32)
        assert self.feature_dim == features.shape[1]
        assert features.shape[0] == labels.shape[0]
        for label, rows in _row_groups(labels, self.bucket_start, self.bucket_num):
            cur = features[rows].astype(np.float64)
            n = cur.shape[0]
            mean = cur.mean(0).astype(np.float32)
            var = cur.var(0, ddof=1 if n != 1 else 0).astype(np.float32)     # fds.py:102
            b = int(F32(label) - F32(self.bucket_start))
            self.num_samples_tracked[b] += n                                    # fds.py:104
            factor = self.momentum if self.momentum is not None else \
                (1 - n / float(self.num_samples_tracked[b]))                    # fds.py:105-106
            factor = 0 if epoch == self.start_update else factor               # fds.py:107
            a, f = F32(1 - factor), F32(factor)
            self.running_mean[b] = a * mean + f * self.running_mean[b]         # fds.py:108-109
            self.running_var[b] = a * var + f * self.running_var[b]            # fds.py:110-111

    def smooth(self, features, labels, epoch):                 # fds.py:115-144
        """In place on ``features`` (A.2); returns it. labels: [B,1]."""
        if epoch < self.start_smooth:
            return features
        labels = np.asarray(labels, dtype=np.float32)
        assert labels.ndim == 2 and labels.shape[1] == 1       # squeeze(1), fds.py:119
        labels = labels[:, 0]
        for label, rows in _row_groups(labels, self.bucket_start, self.bucket_num):
            b = int(F32(label) - F32(self.bucket_start))
            features[rows] = calibrate_mean_var(
                features[rows], self.running_mean_last_epoch[b], self.running_var_last_epoch[b],
                self.smoothed_mean_last_epoch[b], self.smoothed_var_last_epoch[b])
        return features

    def smooth_grad(self, grad_out, labels, epoch):
        """d loss / d features(in) given d loss / d features(out): ``dy * sqrt(factor)`` on the
        calibrated elements, ``dy`` elsewhere (autograd of fds.py:124-143)."""
        gin = np.array(grad_out, dtype=np.float32, copy=True)
        if epoch < self.start_smooth:
            return gin
        labels = np.asarray(labels, dtype=np.float32)[:, 0]
        for label, rows in _row_groups(labels, self.bucket_start, self.bucket_num):
            b = int(F32(label) - F32(self.bucket_start))
            s = calibrate_scale(self.running_var_last_epoch[b], self.smoothed_var_last_epoch[b])
            mult = np.where(s < 0, F32(1), s).astype(np.float32)
            gin[rows] = grad_out[rows] * mult
        return gin
