"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else): CPU restatement of the training-time image transform of
/root/reference/imdb-wiki-dir/datasets.py:38-53 (identical in agedb-dir/datasets.py:38-53) behind ``Resize``:

    transforms.RandomCrop(img_size, padding=16) -> transforms.RandomHorizontalFlip() -> transforms.ToTensor()
    -> transforms.Normalize([.5, .5, .5], [.5, .5, .5])                                   (train split, datasets.py:40-46)
    transforms.ToTensor() -> transforms.Normalize(...)                                     (other splits, datasets.py:48-52)

torchvision is a third-party dependency that /root/reference does not vendor or pin (README: "PyTorch (>= 1.2, tested on
1.6)", i.e. torchvision 0.4 ... 0.7) and that this image does not ship. Its published algorithm, restated here on uint8 HWC
arrays:
  * RandomCrop(size, padding=p): ``F.pad(img, p, fill=0, padding_mode='constant')`` (PIL ``ImageOps.expand(border=p, fill=0)``),
    then ``get_params``: ``i = randint(0, h - th)`` (top), ``j = randint(0, w - tw)`` (left), both inclusive, i drawn first;
    ``F.crop(img, i, j, th, tw)`` = rows i .. i+th-1, columns j .. j+tw-1.
  * RandomHorizontalFlip(p=0.5): one uniform draw AFTER the crop draws; flip (``Image.FLIP_LEFT_RIGHT``) iff it is < 0.5.
  * ToTensor: HWC uint8 -> CHW float32, ``.div(255)``.   * Normalize(mean, std): ``(t - mean) / std`` in float32.
The draws are inputs here (``params[b] = (top, left, flip)``). Pinned in tests/test_augment.py against the PIL operations
torchvision <= 0.7 executes (ImageOps.expand / Image.crop / Image.transpose) and torch's own float32 ops for ToTensor /
Normalize; the random-number STREAM of torchvision is not reproduced (it differs between its own versions: ``random`` up to
0.7, ``torch`` from 0.8), only the order and ranges of the draws."""
import numpy as np

F32 = np.float32


def augment(img_u8, params=None, pad=16):
    """img_u8 [B, S, S, 3] uint8; params [B, 3] int (top, left, flip) or None (evaluation transform) -> [B, 3, S, S] float32."""
    img_u8 = np.asarray(img_u8, dtype=np.uint8)
    b, s = img_u8.shape[0], img_u8.shape[1]
    out = np.empty((b, 3, s, s), dtype=np.float32)
    for k in range(b):
        a = img_u8[k]
        if params is not None:
            top, left, flip = (int(v) for v in params[k])
            assert 0 <= top <= 2 * pad and 0 <= left <= 2 * pad
            a = np.pad(a, ((pad, pad), (pad, pad), (0, 0)))                     # datasets.py:42 (padding=16, fill 0)
            a = a[top:top + s, left:left + s]
            if flip:
                a = a[:, ::-1]                                                  # datasets.py:43
        t = a.transpose(2, 0, 1).astype(F32) / F32(255)                         # ToTensor, datasets.py:44 / :50
        out[k] = (t - F32(0.5)) / F32(0.5)                                      # Normalize, datasets.py:45 / :51
    return out
