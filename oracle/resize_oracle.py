"""TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's checker legs — never by the product).

CPU restatement of the ``transforms.Resize((S, S))`` at the head of the reference's transform chain
(``imdb-wiki-dir/datasets.py:41,49``, ``agedb-dir/datasets.py:41,49``) for a uint8 RGB image. torchvision's Resize on a PIL image is
``img.resize((S, S), Image.BILINEAR)``; the arithmetic lives in the third-party dependency Pillow (absent from /root/reference; the
image pins Pillow 12.2.0), ``src/libImaging/Resample.c``, restated here from its published algorithm:

  * ``precompute_coeffs``: per output index ``xx`` the support window ``[xmin, xmin + n)`` and the triangle-filter weights in float64
    — ``scale = in / out``, ``filterscale = max(scale, 1)`` (antialias when shrinking), ``support = filterscale``,
    ``center = (xx + 0.5) * scale``, ``xmin = max(0, int(center - support + 0.5))``, ``xmax = min(in, int(center + support + 0.5))``,
    ``w[x] = tri((x + xmin - center + 0.5) / filterscale)``, normalised by their SEQUENTIAL float64 sum;
  * ``normalize_coeffs_8bpc``: fixed point with PRECISION_BITS = 32 - 8 - 2 = 22 fractional bits, ``int(0.5 + w * 2^22)``
    (``-0.5`` for negative weights; the triangle filter has none);
  * ``ImagingResampleHorizontal_8bpc`` then ``ImagingResampleVertical_8bpc``: int32 accumulators starting at ``1 << 21``,
    ``clip8(acc >> 22)``; the horizontal pass runs first and ITS RESULT IS ROUNDED TO uint8 before the vertical pass; a pass whose
    size does not change is skipped altogether.

Pinned by tests/test_resize.py against Pillow itself (the library the reference executes) on every shape class: shrink, enlarge,
one axis unchanged, tiny and very elongated images."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size, out_size):
    """(bounds int32 [out][2] = (xmin, n), kk int32 [out][ksize]) of one axis; float64 arithmetic in Pillow's order."""
    scale = np.float64(in_size) / np.float64(out_size)
    filterscale = scale if scale > 1.0 else np.float64(1.0)
    support = np.float64(1.0) * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = np.float64(1.0) / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = (center - support + 0.5).astype(np.int64)               # C cast: truncation toward zero (operands here are > -1)
    xmin = np.maximum(xmin, 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    a = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(a < 1.0, 1.0 - a, 0.0)
    w = np.where(np.arange(ksize)[None, :] < n[:, None], w, 0.0)
    ww = np.zeros(out_size, np.float64)
    for j in range(ksize):                                          # sequential float64 sum, as the C loop
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = (0.5 + w * np.float64(1 << PRECISION_BITS)).astype(np.int64).astype(np.int32)      # weights are >= 0
    return np.stack([xmin, n], 1).astype(np.int32), kk


def _pass(img, bounds, kk, axis):
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for i in range(bounds.shape[0]):
        lo, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = np.tensordot(kk[i, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C], bit for bit what ``PIL.Image.fromarray(img).resize((out_w, out_h), BILINEAR)`` returns."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    if out_w != w:
        img = _pass(img, *coeffs(w, out_w), axis=1)
    if out_h != h:
        img = _pass(img, *coeffs(h, out_h), axis=0)
    return img
