"""Datasets — drop-in for ``imdb-wiki-dir/datasets.py`` / ``agedb-dir/datasets.py`` (``IMDBWIKI`` / ``AgeDB``).

Same constructor and ``__getitem__ -> (img f32[3,S,S], label f32[1], weight f32[1])`` contract; ``.weights`` comes
from the native LDS routine (``dirhip.lds.prepare_weights`` -> ``dir_lds_weights``, bit-exact with the reference's
numpy/scipy arithmetic). Image decoding is host-side PIL (the reference uses torchvision transforms, which this image does not
ship; no GPU JPEG decoder either); Resize is host-side PIL by default and, with ``raw="decoded"`` + ``DeviceResize``, three HIP launches
on the ragged uint8 batch (``dir_resize_u8``: Pillow's bilinear arithmetic bit for bit); the rest of the transform chain — RandomCrop(pad 16) -> RandomHorizontalFlip ->
[0,1] -> Normalize(.5, .5) (datasets.py:38-53) — runs either on the host (numpy, the default ``__getitem__`` contract) or,
with ``raw=True`` + ``DeviceAugment``, as ONE HIP launch per batch on the uint8 images (``dir_augment_u8``):
a quarter of the host->device bytes, no float32 round trip, bf16 NHWC output straight into the MFMA stem. ``SyntheticAgeDataset`` produces device-resident random batches with the
same label/weight semantics for benchmarking without image files.
"""
import logging
import os

import numpy as np
import torch
from torch.utils import data

from .lds import prepare_weights

print = logging.info


def draw_augment_params(n, pad=16, generator=None):
    """The random draws of RandomCrop(S, padding=pad) + RandomHorizontalFlip for ``n`` images: top and left uniform on [0, 2 pad],
    an independent fair flip coin each — int32 ``[n, 3]`` = (top, left, flip) on the host. A single image (the host transform) draws
    in torchvision's order; a batch (DeviceAugment) draws with two vectorised calls (B = 256 used to cost 768 tiny RNG calls on the
    training thread) — same distributions, another stream (torchvision's could not be reproduced across loader workers anyway)."""
    out = torch.empty((n, 3), dtype=torch.int32)
    if n == 1:                                                    # the per-image host transform: torchvision's own order of draws
        out[0, 0] = int(torch.randint(0, 2 * pad + 1, (1,), generator=generator))      # RandomCrop.get_params: i (top) ...
        out[0, 1] = int(torch.randint(0, 2 * pad + 1, (1,), generator=generator))      # ... then j (left)
        out[0, 2] = int(float(torch.rand(1, generator=generator)) < 0.5)               # RandomHorizontalFlip's coin
    elif n:
        out[:, :2] = torch.randint(0, 2 * pad + 1, (n, 2), generator=generator, dtype=torch.int32)
        out[:, 2] = (torch.rand(n, generator=generator) < 0.5).to(torch.int32)
    return out


class DeviceAugment:
    """GPU side of the transform chain for ``raw=True`` datasets: ``aug(u8)`` takes a uint8 ``[B, S, S, 3]`` device batch (decoded,
    resized) and returns the network input, a channels_last ``[B, 3, S, S]`` tensor (float32, or bf16 for the autocast path),
    augmented like datasets.py:40-46 when ``train`` (draws from ``draw_augment_params``; pass ``params`` to replay) and like
    datasets.py:48-52 otherwise. One HIP launch; no CPU fallback."""

    def __init__(self, img_size, train=True, pad=16, dtype=torch.float32, generator=None):
        assert dtype in (torch.float32, torch.bfloat16)
        self.img_size, self.train, self.pad, self.dtype, self.generator = img_size, train, pad, dtype, generator

    def __call__(self, u8, params=None):
        from . import _lib as L
        if not u8.is_cuda:
            raise L.DirHipError(f"DeviceAugment: batch on {u8.device}; the augmentation kernel runs only on the GPU")
        assert u8.dtype == torch.uint8 and u8.dim() == 4 and u8.shape[1] == u8.shape[2] == self.img_size and u8.shape[3] == 3
        u8 = u8.contiguous()
        b, s = u8.shape[0], u8.shape[1]
        if self.train and params is None:
            params = draw_augment_params(b, self.pad, self.generator)
        if params is not None:
            params = torch.as_tensor(params, dtype=torch.int32).to(u8.device).contiguous()
            assert params.shape == (b, 3)
        out = torch.empty((b, s, s, 3), dtype=self.dtype, device=u8.device)
        L.check(L.lib().dir_augment_u8(L.ptr(u8), L.ptr(params), L.ptr(out), L.DIR_BF16 if self.dtype == torch.bfloat16 else L.DIR_F32,
                                       b, s, self.pad, L.stream_ptr(u8.device)), "dir_augment_u8")
        return out.permute(0, 3, 1, 2)                          # [B, 3, S, S] view with channels_last strides


def ragged_collate(samples):
    """collate_fn for ``raw="decoded"`` datasets: images of different sizes cannot be stacked, so a batch is
    ``(flat uint8 [sum H*W*3], sizes int64 [B, 2] = (H, W), labels float32 [B, 1], weights float32 [B, 1][, further per-sample items])`` —
    ``flat`` and ``sizes`` are the input of ``DeviceResize``."""
    imgs = [s[0] for s in samples]
    flat = torch.cat([im.reshape(-1) for im in imgs])
    sizes = torch.tensor([[im.shape[0], im.shape[1]] for im in imgs], dtype=torch.int64)
    labels = torch.from_numpy(np.stack([np.asarray(s[1], dtype=np.float32) for s in samples]))
    weights = torch.from_numpy(np.stack([np.asarray(s[2], dtype=np.float32) for s in samples]))
    extra = tuple(torch.as_tensor([s[i] for s in samples]) for i in range(3, len(samples[0])))      # e.g. the shard-padding flag
    return (flat, sizes, labels, weights) + extra


class DeviceResize:
    """GPU side of ``transforms.Resize((S, S))`` (datasets.py:41,49) for ``raw="decoded"`` batches: ``resize(flat, sizes)`` takes the
    ragged uint8 batch of ``ragged_collate`` (host or device tensors; the copy to the device happens here, non-blocking when pinned) and
    returns the uint8 ``[B, S, S, 3]`` device batch ``DeviceAugment`` consumes. Three HIP launches (``dir_resize_u8``), bit-identical to
    Pillow's bilinear resize; no CPU fallback."""

    def __init__(self, img_size, device=None, stager=None):
        self.img_size = int(img_size)
        self.device = device
        self.stager = stager                                        # measurement hook (tools/pinned_stager.py): an object with to_device(tensor, device); None in the product

    def __call__(self, flat, sizes):
        from . import _lib as L
        dev = torch.device(self.device) if self.device is not None else (flat.device if flat.is_cuda else torch.device("cuda", torch.cuda.current_device()))
        if dev.type != "cuda":
            raise L.DirHipError(f"DeviceResize: target device {dev}; the resize kernels run only on the GPU")
        s = self.img_size
        sizes = torch.as_tensor(sizes, dtype=torch.int64).cpu()
        assert flat.dtype == torch.uint8 and flat.dim() == 1 and sizes.dim() == 2 and sizes.shape[1] == 2
        b = sizes.shape[0]
        h, w = sizes[:, 0], sizes[:, 1]
        nbytes = h * w * 3
        assert int(nbytes.sum()) == flat.numel() and int(h.min()) > 0 and int(w.min()) > 0
        src_off = torch.cumsum(nbytes, 0) - nbytes
        tmp = (h * s * 3 + 15) // 16 * 16                           # intermediate [H][S][3] of every image, 16-byte aligned
        tmp_off = torch.cumsum(tmp, 0) - tmp
        table = torch.stack([src_off, h, w, tmp_off], 1).contiguous()
        lib = L.lib()
        kmax = max(lib.dir_resize_ksize(int(h.max()), s), lib.dir_resize_ksize(int(w.max()), s))
        ws_bytes = lib.dir_resize_u8_workspace(b, s, kmax, int(tmp.sum()))
        flat_d = self.stager.to_device(flat, dev) if self.stager is not None else flat.to(dev, non_blocking=True)
        table_d = table.to(dev, non_blocking=True)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        out = torch.empty((b, s, s, 3), dtype=torch.uint8, device=dev)
        L.check(lib.dir_resize_u8(L.ptr(flat_d), flat_d.numel(), L.ptr(table_d), L.ptr(out), b, s, int(h.max()), kmax, L.ptr(ws), ws_bytes, L.stream_ptr(dev)),
                "dir_resize_u8")
        return out


class DevicePrefetcher:
    """The device half of the input pipeline, ahead of the training loop: a background thread takes host batches from ``batches`` and runs
    ``fn(batch) -> tuple`` (host-to-device copies, ``DeviceResize``, ``DeviceAugment``) on a SIDE stream; the consumer's stream only waits for
    the batch's event. Why a thread and a stream: a copy out of pageable host memory (the ragged decode-only batches) is stream-ordered AND
    blocks its caller — issued on the training stream it makes the host wait for every kernel already queued there, and the loop that should
    run a step ahead of the GPU runs in lockstep with it (measured: 4.8 k instead of 10.3 k img/s, bench.py ``input_pipeline.end_to_end``).
    ``depth`` batches are kept ready. Tensors are handed over with ``record_stream`` (they were allocated on the side stream). The order of
    batches — and of the augmentation draws inside ``fn`` — is that of ``batches``: one producer, sequential."""

    _END = object()

    def __init__(self, batches, device, fn, depth=2):
        import queue
        import threading
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.fn = fn
        self.q = queue.Queue(maxsize=max(1, int(depth)))
        self.stop = threading.Event()
        self.side = torch.cuda.Stream(self.device)
        self.thread = threading.Thread(target=self._run, args=(iter(batches),), daemon=True)
        self.thread.start()

    def _put(self, item):
        import queue
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self, it):
        try:
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.side):
                for batch in it:
                    out = self.fn(batch)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    if not self._put((out, ev)):
                        return
            self._put((self._END, None))
        except BaseException as e:                              # noqa: BLE001  (handed to the consumer)
            self._put((e, None))

    def __iter__(self):
        while True:
            out, ev = self.q.get()
            if out is self._END:
                return
            if isinstance(out, BaseException):
                raise out
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in out:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
            yield out

    def close(self):
        """Stop the producer and WAIT for it: after an early exit of the consumer (``--max_steps``, an exception in the loop) the thread may still be
        inside ``fn(batch)`` — issuing work on the side stream, writing the HBM cache — while the next pass starts. The queue is
        drained so that a producer blocked in ``put`` returns."""
        import queue
        self.stop.set()
        while self.thread.is_alive():
            try:
                self.q.get_nowait()
            except queue.Empty:
                pass
            self.thread.join(timeout=0.05)
        self.side.synchronize()                     # whatever the producer queued on its stream has finished before anybody reuses its buffers

    def __del__(self):
        self.stop.set()


class DeviceImageCache:
    """The RESIZED uint8 training images, resident in HBM: ``[N, S, S, 3]`` bytes = 28.8 GB for IMDB-WIKI-DIR's 191 509 training images at
    224 x 224 — a tenth of one MI355X's 288 GB. ``train.py`` reads the training set TWICE per epoch (the training pass and the FDS
    feature pass, train.py:246-250 / 269-281), both through JPEG decode + Resize on the host; decode is the part of the input pipeline no
    GPU kernel of this image replaces, and two passes of it (2 x 10.3 k img/s) are more than one box's loader delivers (bench.py
    ``input_pipeline.end_to_end``). Resize is deterministic and everything random (crop, flip) comes after it, so the first pass over a sample
    stores its resized bytes here and every later pass — the same epoch's feature pass, every later epoch — gathers them and draws a FRESH
    augmentation on the GPU (``DeviceAugment``): the same distribution of network inputs as the reference's loader, no decode after epoch 0."""

    def __init__(self, n, img_size, device, max_bytes=128 << 30):
        # (a CONTAINER: it computes nothing — the augmentation it hands its batches to is the GPU kernel; a host device only serves the CPU tests
        #  of its index / order / coverage logic)
        self.n, self.s, self.device = int(n), int(img_size), torch.device(device)
        nbytes = self.n * self.s * self.s * 3
        if nbytes > max_bytes:
            raise ValueError(f"DeviceImageCache: {nbytes / 2 ** 30:.1f} GiB for {n} images of {img_size}^2 exceeds the {max_bytes / 2 ** 30:.0f} GiB budget")
        self.u8 = torch.empty((self.n, self.s, self.s, 3), dtype=torch.uint8, device=self.device)
        self.labels = torch.zeros((self.n, 1), dtype=torch.float32, device=self.device)
        self.weights = torch.ones((self.n, 1), dtype=torch.float32, device=self.device)
        self.have = np.zeros(self.n, dtype=bool)                     # host side: which samples have been stored
        self._stored = None                                          # event behind the latest put (puts run on the prefetcher's side stream)

    def put(self, index, u8, labels, weights):
        """``index``: host int64 ``[b]``; ``u8``: device ``[b, S, S, 3]`` (resized, not augmented); labels / weights: device ``[b, 1]``."""
        idx_h = torch.as_tensor(index, dtype=torch.int64).reshape(-1)
        idx_d = idx_h.to(self.device, non_blocking=True)
        self.u8.index_copy_(0, idx_d, u8)
        self.labels.index_copy_(0, idx_d, labels.reshape(-1, 1).float())
        self.weights.index_copy_(0, idx_d, weights.reshape(-1, 1).float())
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._stored = ev
        self.have[idx_h.numpy()] = True

    def covers(self, index):
        return bool(self.have[np.asarray(index, dtype=np.int64)].all())

    def batches(self, index, batch_size, augment, valid=None, shuffle=True, generator=None):
        """One pass over the samples ``index`` (a shard of the epoch), ``batch_size`` at a time, in a fresh random order: yields what the
        loader path yields — ``(network input, labels, weights[, valid])`` on the device."""
        index = torch.as_tensor(index, dtype=torch.int64)
        if self._stored is not None:                                 # a pass may start while the last stores are still queued on another stream
            torch.cuda.current_stream(self.device).wait_event(self._stored)
        order = torch.randperm(len(index), generator=generator) if shuffle else torch.arange(len(index))
        valid_t = None if valid is None else torch.as_tensor(valid, dtype=torch.bool)
        for s0 in range(0, len(index), batch_size):
            sel = order[s0:s0 + batch_size]
            idx_d = index[sel].to(self.device, non_blocking=True)
            x = augment(self.u8.index_select(0, idx_d))
            out = (x, self.labels.index_select(0, idx_d), self.weights.index_select(0, idx_d))
            yield out + ((valid_t[sel],) if valid_t is not None else ())


class _AgeDataset(data.Dataset):
    def __init__(self, df, data_dir, img_size, split='train', reweight='none',
                 lds=False, lds_kernel='gaussian', lds_ks=5, lds_sigma=2, raw=False):
        # raw=True (extension): __getitem__ returns the decoded, resized uint8 HWC image instead of the transformed float
        # tensor; the rest of the transform chain then runs on the GPU (DeviceAugment). raw="decoded": decoded only, at the file's
        # own size — Resize runs on the GPU too (DeviceResize; batches through ragged_collate)
        self.raw = raw
        self.df = df
        self.data_dir = data_dir
        self.img_size = img_size
        self.split = split
        self.weights = self._prepare_weights(reweight=reweight, lds=lds, lds_kernel=lds_kernel, lds_ks=lds_ks,
                                             lds_sigma=lds_sigma)

    def __len__(self):
        return len(self.df)

    def _prepare_weights(self, reweight, max_target=121, lds=False, lds_kernel='gaussian', lds_ks=5, lds_sigma=2):
        return prepare_weights(self.df['age'].values, reweight, max_target=max_target, lds=lds,
                               lds_kernel=lds_kernel, lds_ks=lds_ks, lds_sigma=lds_sigma)

    def get_transform(self):
        size, train = self.img_size, self.split == 'train'

        def transform(img):
            from PIL import Image
            if self.raw == "decoded":
                # decode only: the image leaves the worker at its file size (uint8 [H, W, 3]); Resize runs on the GPU (DeviceResize,
                # dir_resize_u8: Pillow's bilinear arithmetic bit for bit) in front of DeviceAugment — batches need `ragged_collate`
                return torch.from_numpy(np.array(np.asarray(img, dtype=np.uint8), copy=True))
            img = img.resize((size, size), Image.BILINEAR)
            arr = np.asarray(img, dtype=np.uint8)
            if self.raw:
                return torch.from_numpy(np.array(arr, copy=True))              # uint8 [S, S, 3]; augmentation on the GPU
            if train:
                top, left, flip = (int(v) for v in draw_augment_params(1)[0])
                arr = np.pad(arr, ((16, 16), (16, 16), (0, 0)))                 # RandomCrop(size, padding=16)
                arr = arr[top:top + size, left:left + size]
                if flip:                                                         # RandomHorizontalFlip
                    arr = arr[:, ::-1]
            out = torch.from_numpy(np.array(arr, copy=True)).permute(2, 0, 1).float().div_(255.)
            return out.sub_(0.5).div_(0.5)                                       # Normalize([.5]*3, [.5]*3)
        return transform

    def __getitem__(self, index):
        from PIL import Image
        index = index % len(self.df)
        row = self.df.iloc[index]
        img = Image.open(os.path.join(self.data_dir, row['path'])).convert('RGB')
        img = self.get_transform()(img)
        label = np.asarray([row['age']]).astype('float32')
        weight = np.asarray([self.weights[index]]).astype('float32') if self.weights is not None else \
            np.asarray([np.float32(1.)])
        return img, label, weight


class IMDBWIKI(_AgeDataset):
    pass


class AgeDB(_AgeDataset):
    pass


class SyntheticAgeDataset(data.Dataset):
    """Random 224x224 'images' with a given label list; weights from the same LDS routine. Indexable like the
    real datasets (CPU tensors), and ``device_batches`` yields whole batches generated directly in HBM."""

    def __init__(self, labels, img_size=224, reweight='none', lds=False, lds_kernel='gaussian', lds_ks=5,
                 lds_sigma=2, seed=0):
        self.labels = np.asarray(labels, dtype=np.float32)
        self.img_size = img_size
        self.seed = seed
        self.weights = prepare_weights(self.labels, reweight, lds=lds, lds_kernel=lds_kernel, lds_ks=lds_ks,
                                       lds_sigma=lds_sigma)

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(index))
        img = torch.randn(3, self.img_size, self.img_size, generator=g)
        w = np.float32(1.) if self.weights is None else self.weights[index]
        return img, np.asarray([self.labels[index]], dtype=np.float32), np.asarray([w], dtype=np.float32)

    def device_batches(self, indices, batch_size, device, channels_last=True, seed=0, valid=None):
        """Yields ``(x, y, w)`` device batches — ``(x, y, w, valid)`` with the host bool slice when ``valid`` (the padding
        mask of ``parallel.shard_indices``) is given."""
        w_all = None if self.weights is None else torch.as_tensor(np.asarray(self.weights, dtype=np.float32))
        lab_all = torch.as_tensor(self.labels)
        g = torch.Generator(device=device).manual_seed(self.seed * 7919 + seed)
        for s in range(0, len(indices), batch_size):
            idx = indices[s:s + batch_size]
            x = torch.randn(len(idx), 3, self.img_size, self.img_size, device=device, generator=g)
            if channels_last:
                x = x.contiguous(memory_format=torch.channels_last)
            y = lab_all[idx].to(device).view(-1, 1)
            w = torch.ones_like(y) if w_all is None else w_all[idx].to(device).view(-1, 1)
            yield (x, y, w) if valid is None else (x, y, w, valid[s:s + batch_size])
