"""the training-time image augmentation (imdb-wiki-dir/datasets.py:38-53 behind Resize).
CPU: the oracle against the PIL operations torchvision <= 0.7 executes and torch's float32 ToTensor / Normalize arithmetic; the
order and range of the random draws; the raw (uint8) dataset path. -m gpu: dir_augment_u8 against the oracle, bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import augment_oracle


def _pil_chain(arr, top, left, flip, pad=16):
    """What the reference's Compose does to a PIL image (torchvision.transforms.functional on PIL), then ToTensor / Normalize
    with torch's own float32 ops."""
    from PIL import Image, ImageOps
    img = Image.fromarray(arr)
    s = arr.shape[0]
    if top is not None:
        img = ImageOps.expand(img, border=pad, fill=0)                       # F.pad(img, 16, fill=0, 'constant')
        img = img.crop((left, top, left + s, top + s))                      # F.crop(img, i, j, h, w)
        if flip:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)                      # F.hflip
    t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)   # F.to_tensor
    mean = torch.tensor([.5, .5, .5]).view(3, 1, 1)
    std = torch.tensor([.5, .5, .5]).view(3, 1, 1)
    return t.sub_(mean).div_(std).numpy()                                   # F.normalize


def _cases(rng, b, s):
    img = rng.integers(0, 256, (b, s, s, 3), dtype=np.uint8)
    img[0] = 0
    img[1] = 255
    params = np.stack([rng.integers(0, 33, b), rng.integers(0, 33, b), rng.integers(0, 2, b)], 1).astype(np.int32)
    params[:4] = [(0, 0, 0), (32, 32, 1), (0, 32, 1), (32, 0, 0)]
    return img, params


def test_oracle_equals_pil_and_torch_ops():
    rng = np.random.default_rng(0)
    for s in (224, 64, 33):
        img, params = _cases(rng, 6, s)
        got = augment_oracle.augment(img, params)
        for k in range(img.shape[0]):
            want = _pil_chain(img[k], int(params[k, 0]), int(params[k, 1]), int(params[k, 2]))
            assert np.array_equal(got[k], want), (s, k)
        got_eval = augment_oracle.augment(img, None)
        for k in range(img.shape[0]):
            assert np.array_equal(got_eval[k], _pil_chain(img[k], None, None, None))
    assert augment_oracle.augment(np.zeros((1, 8, 8, 3), np.uint8), np.array([[0, 0, 0]]))[0, :, 0, 0].tolist() == [-1.0, -1.0, -1.0]


def test_draw_order_and_ranges():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "imbalanced-regression_amd"))
    from dirhip.datasets import draw_augment_params
    torch.manual_seed(7)
    one = [draw_augment_params(1)[0].tolist() for _ in range(500)]           # the host transform: one image at a time
    torch.manual_seed(7)
    want = []
    for _ in range(500):                                                     # RandomCrop.get_params: i (top) then j (left); then the flip coin
        i = int(torch.randint(0, 33, (1,)))
        j = int(torch.randint(0, 33, (1,)))
        want.append((i, j, int(float(torch.rand(1)) < 0.5)))
    assert one == [list(w) for w in want]
    p = draw_augment_params(20000)                                           # a batch: vectorised draws, same distributions
    assert p[:, :2].min() == 0 and p[:, :2].max() == 32 and set(p[:, 2].tolist()) == {0, 1}
    assert abs(p[:, 0].float().mean() - 16) < 0.3 and abs(p[:, 1].float().mean() - 16) < 0.3 and abs(p[:, 2].float().mean() - 0.5) < 0.02
    assert abs(float(torch.corrcoef(p[:, :2].float().T)[0, 1])) < 0.03
    g = torch.Generator().manual_seed(3)
    a = draw_augment_params(10, generator=g)
    g = torch.Generator().manual_seed(3)
    assert torch.equal(a, draw_augment_params(10, generator=g))


def test_raw_dataset_matches_host_transform(tmp_path):
    import sys
    import pandas as pd
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "imbalanced-regression_amd"))
    from dirhip.datasets import AgeDB, draw_augment_params
    rng = np.random.default_rng(1)
    rows = []
    for k in range(3):
        arr = rng.integers(0, 256, (70 + 5 * k, 50 + 9 * k, 3), dtype=np.uint8)
        Image.fromarray(arr).save(tmp_path / f"im{k}.png")
        rows.append({"path": f"im{k}.png", "age": 20 + k, "split": "train"})
    df = pd.DataFrame(rows)
    raw = AgeDB(df, str(tmp_path), 64, split="train", raw=True)
    host = AgeDB(df, str(tmp_path), 64, split="train")
    for k in range(3):
        u8, lab, w = raw[k]
        assert u8.dtype == torch.uint8 and tuple(u8.shape) == (64, 64, 3) and lab.tolist() == [20.0 + k]
        torch.manual_seed(11 + k)
        img, _, _ = host[k]                                                  # host path: draws one (top, left, flip)
        torch.manual_seed(11 + k)
        params = draw_augment_params(1)
        want = augment_oracle.augment(u8.numpy()[None], params.numpy())[0]
        assert np.array_equal(img.numpy(), want)
    ev = AgeDB(df, str(tmp_path), 64, split="val")
    img, _, _ = ev[1]
    assert np.array_equal(img.numpy(), augment_oracle.augment(raw[1][0].numpy()[None], None)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("s,b", [(224, 16), (64, 5), (33, 3)])
def test_device_augment_bit_equal_oracle(s, b):
    from dirhip.datasets import DeviceAugment, draw_augment_params
    rng = np.random.default_rng(s)
    img, params = _cases(rng, max(b, 4), s)
    u8 = torch.as_tensor(img).cuda()
    aug = DeviceAugment(s, train=True)
    y = aug(u8, params=torch.as_tensor(params))
    assert y.shape == (img.shape[0], 3, s, s) and y.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.float32
    assert np.array_equal(y.cpu().numpy(), augment_oracle.augment(img, params))
    y16 = DeviceAugment(s, train=True, dtype=torch.bfloat16)(u8, params=torch.as_tensor(params))
    assert torch.equal(y16, y.to(torch.bfloat16)) and y16.is_contiguous(memory_format=torch.channels_last)
    ye = DeviceAugment(s, train=False)(u8)
    assert np.array_equal(ye.cpu().numpy(), augment_oracle.augment(img, None))
    # drawing inside: reproducible from the generator, every image within the documented ranges
    g = torch.Generator().manual_seed(5)
    y1 = DeviceAugment(s, train=True, generator=g)(u8)
    g = torch.Generator().manual_seed(5)
    want = augment_oracle.augment(img, draw_augment_params(img.shape[0], generator=g).numpy())
    assert np.array_equal(y1.cpu().numpy(), want)
    with pytest.raises(Exception):
        aug(torch.as_tensor(img))                                            # host tensor: no CPU fallback


@pytest.mark.gpu
def test_device_augment_feeds_the_network():
    """uint8 batch -> DeviceAugment(bf16) -> resnet50 forward: the stem takes the channels_last bf16 tensor as it is."""
    from dirhip.datasets import DeviceAugment
    from dirhip.resnet import resnet50
    torch.manual_seed(0)
    model = resnet50(fds=False, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2,
                     momentum=0.9).cuda().to(memory_format=torch.channels_last).eval()
    u8 = torch.randint(0, 256, (4, 224, 224, 3), dtype=torch.uint8, device="cuda")
    x32 = DeviceAugment(224, train=False)(u8)
    x16 = DeviceAugment(224, train=False, dtype=torch.bfloat16)(u8)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a = model(x32)
        b = model(x16)
    assert torch.equal(a, b) and torch.isfinite(a).all()


@pytest.mark.gpu
def test_device_image_cache_serves_the_stored_bytes_with_fresh_augmentation():
    """datasets.DeviceImageCache: what put() stored is what batches() augments — every sample of the shard exactly once per pass, labels and
    weights travelling with their images, `covers` true only once every requested sample is in."""
    import torch
    from dirhip.datasets import DeviceAugment, DeviceImageCache
    dev = torch.device("cuda", 0)
    n, s = 40, 32
    g = torch.Generator().manual_seed(0)
    imgs = torch.randint(0, 256, (n, s, s, 3), dtype=torch.uint8, generator=g)
    labels = torch.arange(n, dtype=torch.float32).view(-1, 1)
    weights = (torch.arange(n, dtype=torch.float32) * 0.5 + 1).view(-1, 1)
    cache = DeviceImageCache(n, s, dev)
    idx_a = torch.tensor([3, 7, 1, 39, 20])
    assert not cache.covers(idx_a)
    cache.put(idx_a, imgs[idx_a].to(dev), labels[idx_a].to(dev), weights[idx_a].to(dev))
    assert cache.covers(idx_a) and not cache.covers(torch.arange(n))
    rest = torch.tensor([i for i in range(n) if i not in set(idx_a.tolist())])
    cache.put(rest, imgs[rest].to(dev), labels[rest].to(dev), weights[rest].to(dev))
    assert cache.covers(torch.arange(n))
    aug = DeviceAugment(s, train=False, dtype=torch.float32)        # eval transform: deterministic, so the pixels can be compared
    shard = torch.tensor([5, 6, 7, 8, 9, 30, 31, 2, 0, 39, 17])
    seen = []
    for x, y, w, valid in cache.batches(shard, 4, aug, valid=[True] * 10 + [False]):
        ids = y.view(-1).long().cpu()
        seen += ids.tolist()
        assert torch.equal(w.cpu().view(-1), weights[ids].view(-1)) and valid.dtype == torch.bool and len(valid) == len(ids)
        want = aug(imgs[ids].to(dev))
        assert torch.equal(x, want)
    assert sorted(seen) == sorted(shard.tolist())
