"""FDS, NYUD2-DIR dense variant — drop-in for ``nyud2-dir/models/fds.py`` (class ``FDS``) and
``nyud2-dir/util.py::calibrate_mean_var``.

Differences from the age variant, all reproduced:
  * features are dense maps ``[B, C, H, W]`` (C = 128 in the reference network), labels ``[B, 1, H, W]`` depth maps;
    every pixel is a row: bucket = ``clamp(int(depth * 10), bucket_start, bucket_num - 1)`` (fds.py:51-53, :138-139),
    defaults ``bucket_num=100, bucket_start=7``;
  * ``smooth`` is NOT in place: it works on an NHWC copy and returns a new ``[B, C, H, W]`` view of it (fds.py:128-149);
  * calibration clip ``[0.2, 5]`` with the ``v1 <= 0`` / ``v2 < 0`` guard as it executes on torch >= 1.2 (see
    ``fds_stsb``; ``per_column_guard=True`` selects the intended semantics);
  * ``update_running_stats`` re-creates ``running_mean`` / ``running_var`` / ``num_samples_tracked`` (the reference moves
    them to the CPU and back, fds.py:88-96,105,126), which BREAKS the alias ``running_*_last_epoch is running_*`` that
    ``_update_last_epoch_stats`` sets up: the last-epoch tables are snapshots (SURVEY A.1 vs Appendix D).
Kernels: ``dir_fds_bin_scaled`` + the shared scatter / finalize / smooth-bins / prepare-scale / calibrate kernels
(their narrow-row paths: a 128-channel row is 512 B, so a workgroup processes 8 rows at a time).
"""
import torch

from . import _lib as L
from . import ops
from .fds import FDS as _AgeFDS
from .fds import merge_stats_across_ranks


def _rows(features):
    """[B, C, H, W] -> contiguous [B*H*W, C] float32 (free when the map is channels_last already)."""
    b, c, h, w = features.shape
    f = features.permute(0, 2, 3, 1)
    if f.dtype != torch.float32:
        f = f.float()
    return f.contiguous().view(-1, c), (b, h, w, c)


def _nchw_ok(t):
    return t.dim() == 4 and t.dtype == torch.float32 and t.is_contiguous() and (t.shape[2] * t.shape[3]) % 4 == 0


class _DenseSmoothNCHWFn(torch.autograd.Function):
    """The calibration on the network's own NCHW map, tables resident in LDS (``dir_fds_calibrate_*_nchw``): no permuted copy of the
    284 MB map before and after (nyud2-dir/models/fds.py:136,149). Returns a NEW tensor (fds.py:136 works on a copy)."""

    @staticmethod
    def forward(ctx, features, bins, m1, scale, m2):
        y = ops.calibrate_nchw(features, bins, m1, scale, m2)
        if y is None:
            raise L.DirHipError("dir_fds_calibrate_fwd_nchw does not take this geometry")
        ctx.save_for_backward(bins, scale)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        bins, scale = ctx.saved_tensors
        g = grad_out if (grad_out.dtype == torch.float32 and grad_out.is_contiguous()) else grad_out.float().contiguous()
        dx = ops.calibrate_bwd_nchw(g, bins, scale)
        if dx is None:
            raise L.DirHipError("dir_fds_calibrate_bwd_nchw does not take this geometry")
        return dx, None, None, None, None


def _nchw_fits(c, nb):
    return 3 * c * ((nb + 3) // 4 * 4) * 4 <= 160 * 1024


class _DenseSmoothFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, bins, m1, scale, m2):
        rows, (b, h, w, c) = _rows(features)
        if rows.data_ptr() == features.data_ptr():
            rows = rows.clone()                               # never touch the caller's tensor (fds.py:136 works on a copy)
        ops.calibrate_fwd_(rows, bins, m1, scale, m2)
        ctx.save_for_backward(bins, scale)
        ctx.shape = (b, h, w, c)
        return rows.view(b, h, w, c).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        bins, scale = ctx.saved_tensors
        b, h, w, c = ctx.shape
        g = grad_out.permute(0, 2, 3, 1).contiguous().view(-1, c)
        if g.dtype != torch.float32:
            g = g.float()
        dx = ops.calibrate_bwd(g, bins, scale)
        return dx.view(b, h, w, c).permute(0, 3, 1, 2), None, None, None, None


class FDS(_AgeFDS):
    CLIP = (0.2, 5.0)
    GUARD_MODE = 1

    def __init__(self, feature_dim, bucket_num=100, bucket_start=7, start_update=0, start_smooth=1,
                 kernel='gaussian', ks=5, sigma=2, momentum=0.9, per_column_guard=False):
        super().__init__(feature_dim, bucket_num, bucket_start, start_update, start_smooth, kernel, ks, sigma, momentum)
        if per_column_guard:
            self.GUARD_MODE = 2

    def _get_bucket_idx(self, label):
        import numpy as np
        label = np.float32(label.cpu() if hasattr(label, "cpu") else label)
        return max(min(int(label * np.float32(10)), self.bucket_num - 1), self.bucket_start)

    def _bins(self, labels):
        labels = labels.reshape(-1)
        labels = L.require_device_tensor((labels if labels.dtype == torch.float32 else labels.float()).contiguous(),
                                         torch.float32, "labels")
        return ops.bin_scaled(labels, 10.0, self.bucket_start, self.bucket_num)

    def local_stats(self, features, labels):
        rows, _ = _rows(features)
        L.require_device_tensor(rows, torch.float32, "features")
        return ops.scatter_stats(rows, self._bins(labels.squeeze(1)), self.bucket_num - self.bucket_start)

    def update_running_stats(self, features, labels, epoch):
        if epoch < int(self.epoch.item()):
            return
        assert self.feature_dim == features.size(1), "Input feature dimension is not aligned!"
        assert features.size(0) == labels.size(0), "Dimensions of features and labels are not aligned!"
        # the reference round-trips these three buffers through the CPU: new tensor objects, alias with *_last_epoch gone
        self.num_samples_tracked = self.num_samples_tracked.clone()
        self.running_mean = self.running_mean.clone()
        self.running_var = self.running_var.clone()
        count, mean, m2 = self.local_stats(features, labels)
        if self._world() > 1:
            count, mean, m2 = merge_stats_across_ranks(count, mean, m2, self.process_group)
        self.apply_stats(count, mean, m2, epoch)

    def smooth(self, features, labels, epoch):
        if epoch < self.start_smooth:
            return features
        if not features.is_cuda:
            raise L.DirHipError(f"features on {features.device}: FDS runs only as HIP kernels on an AMD GPU (no CPU fallback)")
        assert features.dim() == 4 and features.size(1) == self.feature_dim
        fn = _DenseSmoothNCHWFn if (_nchw_ok(features) and _nchw_fits(features.shape[1], self.bucket_num - self.bucket_start)) else _DenseSmoothFn
        return fn.apply(features, self._bins(labels.squeeze(1)), self.running_mean_last_epoch.contiguous(),
                        self._scale_table(), self.smoothed_mean_last_epoch.contiguous())


def calibrate_mean_var(matrix, m1, v1, m2, v2, clip_min=0.2, clip_max=5., per_column_guard=False):
    """nyud2-dir/util.py:151-162 on the GPU (returns a new tensor)."""
    from .fds_stsb import calibrate_mean_var as _cal
    return _cal(matrix, m1, v1, m2, v2, clip_min, clip_max, per_column_guard)
