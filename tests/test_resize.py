"""Resize((S, S)) of the reference's transform chain (imdb-wiki-dir/datasets.py:41,49 = Pillow's bilinear resize, the arithmetic of its
Resample.c) — SURVEY §8f-4.
  * CPU: oracle/resize_oracle.py against Pillow itself (the library the reference executes), every shape class;
  * -m gpu: dir_resize_u8 on ragged batches against the oracle AND against Pillow, bit for bit; the chain Resize -> augment against the
    host chain; edge cases (one image, an axis already at S, 1-pixel images, extreme aspect ratios, upscaling)."""
import os

import numpy as np
import pytest
import torch

from oracle.resize_oracle import coeffs, resize_bilinear_u8

SHAPES = [(320, 320), (500, 375), (224, 224), (100, 150), (1000, 700), (17, 1031), (224, 300), (400, 224), (3, 5), (225, 223), (1, 1), (1, 900),
          (64, 64), (223, 225), (449, 447)]


def _pil(img, s):
    from PIL import Image
    return np.asarray(Image.fromarray(img).resize((s, s), Image.BILINEAR))


def test_oracle_equals_pillow_bit_for_bit():
    rng = np.random.default_rng(0)
    for s in (224, 112):
        for (h, w) in SHAPES:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            assert np.array_equal(resize_bilinear_u8(img, s, s), _pil(img, s)), (h, w, s)
    # structured images: constant (weights must sum to exactly 2^22 after rounding or the value drifts), ramps, extremes
    for val in (0, 1, 127, 254, 255):
        img = np.full((333, 517, 3), val, np.uint8)
        assert np.array_equal(resize_bilinear_u8(img, 224, 224), _pil(img, 224))
    ramp = (np.arange(600)[None, :, None] % 256 * np.ones((450, 1, 3))).astype(np.uint8)
    assert np.array_equal(resize_bilinear_u8(ramp, 224, 224), _pil(ramp, 224))


def test_oracle_coefficients_sanity():
    b, k = coeffs(320, 224)
    assert b.shape == (224, 2) and k.shape[1] == 5                        # ksize = ceil(320 / 224) * 2 + 1
    assert np.all(b[:, 0] >= 0) and np.all(b[:, 0] + b[:, 1] <= 320) and np.all(k >= 0)
    assert np.all(np.abs(k.sum(1) - (1 << 22)) <= 2)                      # normalised weights in 22-bit fixed point
    b, k = coeffs(100, 224)                                               # enlarging: support 1, three taps
    assert k.shape[1] == 3 and b[:, 1].max() <= 3


@pytest.mark.gpu
def test_device_resize_ragged_batch_bit_exact_vs_oracle_and_pillow():
    from dirhip.datasets import DeviceResize
    rng = np.random.default_rng(1)
    for s in (224, 96):
        imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (h, w) in SHAPES]
        flat = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs]))
        sizes = torch.tensor([im.shape[:2] for im in imgs])
        out = DeviceResize(s)(flat, sizes).cpu().numpy()
        assert out.shape == (len(imgs), s, s, 3)
        for i, im in enumerate(imgs):
            assert np.array_equal(out[i], resize_bilinear_u8(im, s, s)), ("oracle", im.shape, s)
            assert np.array_equal(out[i], _pil(im, s)), ("pillow", im.shape, s)
    # one image; a batch of identical sizes (the regular case of a curated dataset); inputs already on the device
    one = rng.integers(0, 256, (301, 199, 3), dtype=np.uint8)
    got = DeviceResize(224)(torch.from_numpy(one.reshape(-1)).cuda(), torch.tensor([[301, 199]])).cpu().numpy()[0]
    assert np.array_equal(got, _pil(one, 224))
    same = rng.integers(0, 256, (32, 320, 320, 3), dtype=np.uint8)
    got = DeviceResize(224)(torch.from_numpy(same.reshape(-1)), torch.tensor([[320, 320]] * 32)).cpu().numpy()
    for i in range(32):
        assert np.array_equal(got[i], _pil(same[i], 224))


@pytest.mark.gpu
def test_decode_only_workers_plus_gpu_resize_and_augment_equal_the_host_resize_chain(tmp_path):
    """Files -> IMDBWIKI(raw="decoded") -> ragged_collate -> DeviceResize -> DeviceAugment == IMDBWIKI(raw=True) (host Pillow resize)
    -> DeviceAugment, with the same crop / flip draws: identical network inputs."""
    import pandas as pd
    from PIL import Image
    from torch.utils.data import DataLoader
    from dirhip.datasets import IMDBWIKI, DeviceAugment, DeviceResize, draw_augment_params, ragged_collate
    rng = np.random.default_rng(2)
    rows = []
    for i, (h, w) in enumerate([(320, 320), (250, 400), (480, 360), (224, 224), (100, 90), (640, 200)]):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        arr = np.stack([128 + 100 * np.sin(xx / (7 + i) + c) * np.cos(yy / (5 + c)) for c in range(3)], -1)
        Image.fromarray(np.clip(arr + rng.normal(0, 10, arr.shape), 0, 255).astype(np.uint8)).save(tmp_path / f"f{i}.png")
        rows.append({"path": f"f{i}.png", "age": float(20 + i), "split": "train"})
    df = pd.DataFrame(rows)
    ds_dec = IMDBWIKI(df, str(tmp_path), img_size=224, split="train", raw="decoded")
    ds_raw = IMDBWIKI(df, str(tmp_path), img_size=224, split="train", raw=True)
    flat, sizes, lab, wts = next(iter(DataLoader(ds_dec, batch_size=6, collate_fn=ragged_collate, num_workers=2)))
    img_raw, lab2, wts2 = next(iter(DataLoader(ds_raw, batch_size=6)))
    assert torch.equal(lab, lab2) and torch.equal(wts, wts2) and sizes.tolist() == [[320, 320], [250, 400], [480, 360], [224, 224], [100, 90], [640, 200]]
    u8 = DeviceResize(224)(flat, sizes)
    assert torch.equal(u8.cpu(), img_raw)                                   # GPU resize == the worker's Pillow resize
    params = draw_augment_params(6, generator=torch.Generator().manual_seed(0))
    a = DeviceAugment(224, train=True, dtype=torch.bfloat16)(u8, params=params)
    b = DeviceAugment(224, train=True, dtype=torch.bfloat16)(img_raw.cuda(), params=params)
    assert torch.equal(a, b)
