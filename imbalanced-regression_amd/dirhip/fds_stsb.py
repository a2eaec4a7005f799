"""FDS, STS-B-DIR variant — drop-in for ``sts-b-dir/fds.py`` (class ``FDS``) and ``sts-b-dir/util.py::calibrate_mean_var``.

Same state, same epoch protocol and aliasing as the age variant (``dirhip.fds.FDS``); what differs:
  * bins: ``_get_bucket_idx`` = position of the label among the ``np.histogram(bins=bucket_num, range=(0, 5))`` edges,
    ``5.0 -> bucket_num - 1``, clamped below by ``bucket_start`` (fds.py:51-57) — no boundary lumping;
  * calibration clip ``[0.5, 2]`` and the ``v1 <= 0`` / ``v2 < 0`` guard (util.py:63-73). NOTE the guard as it EXECUTES
    on torch >= 1.2: ``((v1 > 0.) + (v2 >= 0.)) == 2`` adds bool tensors (logical or) and is never ``== 2``, so a
    bucket with ANY such column is returned unchanged as a whole. ``GUARD_MODE = 1`` reproduces that (parity with the
    reference run in this image); ``FDS(..., per_column_guard=True)`` selects the evident intent (torch 0.4.1 semantics:
    only those columns stay untouched);
  * buckets that received no sample in an update are filled from their neighbours afterwards (fds.py:112-125);
  * defaults ``bucket_num=50``, feature dim 4*2*1500 = 12000 in the reference model.
Kernels: ``dir_fds_bin_edges`` + the shared scatter / finalize / smooth-bins / calibrate kernels +
``dir_fds_fill_empty_buckets`` + ``dir_fds_prepare_scale_ex``.
"""
import numpy as np
import torch

from . import _lib as L
from . import ops
from .fds import FDS as _AgeFDS
from .fds import merge_stats_across_ranks


class _SmoothBinsFn(torch.autograd.Function):
    """In-place calibration given precomputed table rows (fds.py:127-141)."""

    @staticmethod
    def forward(ctx, features, bins, m1, scale, m2):
        ops.calibrate_fwd_(features, bins, m1, scale, m2)
        ctx.mark_dirty(features)
        ctx.save_for_backward(bins, scale)
        return features

    @staticmethod
    def backward(ctx, grad_out):
        bins, scale = ctx.saved_tensors
        return ops.calibrate_bwd(grad_out, bins, scale), None, None, None, None


class FDS(_AgeFDS):
    CLIP = (0.5, 2.0)
    GUARD_MODE = 1

    def __init__(self, feature_dim, bucket_num=50, bucket_start=0, start_update=0, start_smooth=1,
                 kernel='gaussian', ks=5, sigma=2, momentum=0.9, per_column_guard=False):
        super().__init__(feature_dim, bucket_num, bucket_start, start_update, start_smooth, kernel, ks, sigma, momentum)
        if per_column_guard:
            self.GUARD_MODE = 2
        # float32 edges exactly as the reference builds them on every call (fds.py:53)
        _, edges = np.histogram(a=np.array([], dtype=np.float32), bins=bucket_num, range=(0., 5.))
        self._edges_host = torch.tensor(np.asarray(edges, dtype=np.float32))
        self._edges = None

    def _edges_on(self, device):
        if self._edges is None or self._edges.device != device:
            self._edges = self._edges_host.to(device)
        return self._edges

    def _get_bucket_idx(self, label):
        """Host restatement for single labels (API compatibility; the hot path uses dir_fds_bin_edges)."""
        label = np.float32(label)
        edges = self._edges_host.numpy()
        if label == 5.:
            return self.bucket_num - 1
        return max(int(np.where(edges > label)[0][0]) - 1, self.bucket_start)

    def _bins(self, labels):
        labels = labels.reshape(-1)
        labels = L.require_device_tensor((labels if labels.dtype == torch.float32 else labels.float()).contiguous(),
                                         torch.float32, "labels")
        return ops.bin_edges(labels, self._edges_on(labels.device), self.bucket_start, self.bucket_num)

    def local_stats(self, features, labels):
        features = L.require_device_tensor(features if features.dtype == torch.float32 else features.float(),
                                           torch.float32, "features")
        return ops.scatter_stats(features, self._bins(labels), self.bucket_num - self.bucket_start)

    def update_running_stats(self, features, labels, epoch):
        if epoch < int(self.epoch.item()):
            return
        assert self.feature_dim == features.size(1), "Input feature dimension is not aligned!"
        assert features.size(0) == labels.size(0), "Dimensions of features and labels are not aligned!"
        count, mean, m2 = self.local_stats(features, labels)
        if self._world() > 1:
            count, mean, m2 = merge_stats_across_ranks(count, mean, m2, self.process_group)
        self.apply_stats(count, mean, m2, epoch)
        ops.fill_empty_buckets(count, self.running_mean, self.running_var)      # fds.py:112-125
        self._invalidate()

    def _smooth_apply(self, features, labels):
        return _SmoothBinsFn.apply(features, self._bins(labels), self.running_mean_last_epoch, self._scale_table(),
                                   self.smoothed_mean_last_epoch)


class _CalibrateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, matrix, m1, v1, m2, v2, clip_min, clip_max, guard_mode):
        c = matrix.shape[1]
        scale = ops.prepare_scale(v1.reshape(1, c).contiguous(), v2.reshape(1, c).contiguous(), clip_min, clip_max, guard_mode=guard_mode)
        bins = torch.zeros(matrix.shape[0], dtype=torch.int32, device=matrix.device)
        out = matrix.clone(memory_format=torch.contiguous_format)
        ops.calibrate_fwd_(out, bins, m1.reshape(1, c).contiguous(), scale, m2.reshape(1, c).contiguous())
        ctx.save_for_backward(bins, scale)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        bins, scale = ctx.saved_tensors
        return ops.calibrate_bwd(grad_out, bins, scale), None, None, None, None, None, None, None


def calibrate_mean_var(matrix, m1, v1, m2, v2, clip_min=0.5, clip_max=2., per_column_guard=False):
    """sts-b-dir/util.py:63-73 on the GPU (returns a new tensor); guard semantics as in the module docstring."""
    for t, nm in ((matrix, "matrix"), (m1, "m1"), (v1, "v1"), (m2, "m2"), (v2, "v2")):
        L.require_device_tensor(t if t.is_contiguous() else t.contiguous(), torch.float32, nm)
    return _CalibrateFn.apply(matrix, m1, v1, m2, v2, float(clip_min), float(clip_max), 2 if per_column_guard else 1)
