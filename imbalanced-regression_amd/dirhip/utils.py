"""Drop-in for ``imdb-wiki-dir/utils.py`` (= ``agedb-dir/utils.py``).

Kernel-backed: ``calibrate_mean_var`` (utils.py:97-107). Host, bit-exact: ``get_lds_kernel_window``
(utils.py:110-122; scipy on the host like the reference). The remaining helpers
(``AverageMeter`` ... ``save_checkpoint``) are plain host utilities ``train.py`` star-imports, together with
the names ``torch`` / ``np`` / ``os`` / ``logging`` / ``shutil`` it relies on.
"""
import logging
import os
import shutil

import numpy as np
import torch
from scipy.ndimage import gaussian_filter1d
from scipy.signal.windows import triang

from . import _lib as L
from . import ops


class AverageMeter(object):
    """Running value / average of one scalar (utils.py:10-30)."""

    def __init__(self, name, fmt=':f'):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val, self.count = val, self.count + n
        self.sum += val * n
        self.avg = self.sum / self.count

    def __str__(self):
        return ('{name} {val' + self.fmt + '} ({avg' + self.fmt + '})').format(**self.__dict__)


class ProgressMeter(object):
    """utils.py:33-48."""

    def __init__(self, num_batches, meters, prefix=""):
        width = len(str(num_batches // 1))
        self.batch_fmtstr = '[{:' + str(width) + 'd}/' + ('{:' + str(width) + 'd}').format(num_batches) + ']'
        self.meters = meters
        self.prefix = prefix

    def display(self, batch):
        logging.info('\t'.join([self.prefix + self.batch_fmtstr.format(batch)] + [str(m) for m in self.meters]))


def query_yes_no(question):
    """utils.py:51-64: ask on the terminal. Without a terminal (torchrun, batch jobs) nobody can answer, and the
    question guards an ``rmtree``: the answer is then NO — the caller must opt in explicitly (``--overwrite``)."""
    import sys
    if not sys.stdin or not sys.stdin.isatty():
        return False
    answers = {"yes": True, "y": True, "ye": True, "no": False, "n": False}
    while True:
        print(question + " [Y/n] ", end=':')
        choice = input().lower()
        if choice == '':
            return True
        if choice in answers:
            return answers[choice]
        print("Please respond with 'yes' or 'no' (or 'y' or 'n').\n")


def prepare_folders(args):
    """utils.py:67-78. An existing run folder (with its checkpoints) is deleted only on an explicit yes: the
    ``--overwrite`` flag, or the interactive answer when there is a terminal."""
    store = os.path.join(args.store_root, args.store_name)
    if os.path.exists(store) and not args.resume and not args.pretrained and not args.evaluate:
        if getattr(args, "overwrite", False) or query_yes_no('overwrite previous folder: {} ?'.format(store)):
            shutil.rmtree(store)
            print(store + ' removed.')
        else:
            raise RuntimeError('Output folder {} already exists; pass --overwrite to replace it or --resume <ckpt> to '
                               'continue it'.format(store))
    for folder in (args.store_root, store):
        if not os.path.exists(folder):
            print(f"===> Creating folder: {folder}")
            os.makedirs(folder, exist_ok=True)


def adjust_learning_rate(optimizer, epoch, args):
    """utils.py:81-86: step schedule, x0.1 at every milestone reached."""
    lr = args.lr
    for milestone in args.schedule:
        lr *= 0.1 if epoch >= milestone else 1.
    for group in optimizer.param_groups:
        group['lr'] = lr


def save_checkpoint(args, state, is_best, prefix=''):
    """utils.py:89-94: <store_root>/<store_name>/ckpt.pth.tar (+ ckpt.best.pth.tar)."""
    latest = os.path.join(args.store_root, args.store_name, prefix + "ckpt.pth.tar")
    torch.save(state, latest)
    if not is_best:
        return
    logging.info("===> Saving current best checkpoint...")
    shutil.copyfile(latest, latest.replace('pth.tar', 'best.pth.tar'))


class _CalibrateFn(torch.autograd.Function):
    """Out of place (utils.py:106-107) or, ``inplace``, on ``matrix`` itself (utils.py:100-104: ``matrix[:, valid] = ...; return matrix``)."""

    @staticmethod
    def forward(ctx, matrix, m1, m2, scale, inplace):
        c = matrix.shape[1]
        bins = torch.zeros(matrix.shape[0], dtype=torch.int32, device=matrix.device)
        if inplace:
            ctx.mark_dirty(matrix)
            out = matrix
        else:
            out = matrix.clone(memory_format=torch.contiguous_format)
        ops.calibrate_fwd_(out, bins, m1.reshape(1, c).contiguous(), scale, m2.reshape(1, c).contiguous())
        ctx.save_for_backward(bins, scale)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        bins, scale = ctx.saved_tensors
        return ops.calibrate_bwd(grad_out.contiguous(), bins, scale), None, None, None, None


def calibrate_mean_var(matrix, m1, v1, m2, v2, clip_min=0.1, clip_max=10):
    """utils.py:97-107 on the GPU, with the reference's three branches AND their object semantics: ``sum(v1) < 1e-10`` -> the input object, untouched;
    some ``v1 == 0`` -> the input is calibrated IN PLACE on the columns with ``v1 != 0`` and returned itself (autograd sees an in-place operation,
    as in the reference); otherwise a NEW tensor ``(matrix - m1) * sqrt(clamp(v2 / v1)) + m2``. matrix [n, C]; m1, v1, m2, v2 [C]. Which branch it
    is comes from the per-column multipliers ``dir_fds_prepare_scale`` emits (-1 = column untouched) with one host read, where the reference
    syncs twice (``if torch.sum(v1) < 1e-10``, ``if (v1 == 0.).any()``)."""
    for t, nm in ((matrix, "matrix"), (m1, "m1"), (v1, "v1"), (m2, "m2"), (v2, "v2")):
        L.require_device_tensor(t if t.is_contiguous() else t.contiguous(), torch.float32, nm)
    assert matrix.dim() == 2
    c = matrix.shape[1]
    scale = ops.prepare_scale(v1.reshape(1, c).contiguous(), v2.reshape(1, c).contiguous(), float(clip_min), float(clip_max))
    untouched = int((scale < 0).sum().item())
    if untouched == c:                                     # utils.py:98-99 (or every column has v1 == 0: the in-place branch would touch nothing)
        return matrix
    if untouched > 0:                                      # utils.py:100-104
        if not matrix.is_contiguous():
            raise L.DirHipError("calibrate_mean_var: the in-place branch (some v1 == 0) needs a contiguous matrix")
        return _CalibrateFn.apply(matrix, m1, m2, scale, True)
    return _CalibrateFn.apply(matrix, m1, m2, scale, False)


def get_lds_kernel_window(kernel, ks, sigma):
    """utils.py:110-122: float64 numpy window; gaussian / laplace divided by their MAX, triang as is."""
    assert kernel in ['gaussian', 'triang', 'laplace']
    half_ks = (ks - 1) // 2
    if kernel == 'gaussian':
        delta = [0.] * half_ks + [1.] + [0.] * half_ks
        smoothed = gaussian_filter1d(delta, sigma=sigma)
        return smoothed / max(smoothed)
    if kernel == 'triang':
        return triang(ks)
    taps = [np.exp(-abs(x) / sigma) / (2. * sigma) for x in np.arange(-half_ks, half_ks + 1)]
    return np.asarray(taps) / max(taps)
