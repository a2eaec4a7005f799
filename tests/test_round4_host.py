"""CPU-side checks added in round 4 (ADVICE r3 items + bench launcher + host logic). No GPU needed."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adam_fallback_with_closure_takes_the_step_and_returns_the_loss():
    from dirhip.optim import Adam
    p = torch.nn.Parameter(torch.randn(5, 3))                     # CPU parameter: torch's own step
    q = p.detach().clone().requires_grad_(True)
    def closure_for(t):
        def c():
            t.grad = None
            l = (t * t).sum()
            l.backward()
            return l
        return c
    a, b = Adam([p], lr=1e-2), torch.optim.Adam([q], lr=1e-2)
    la, lb = a.step(closure_for(p)), b.step(closure_for(q))
    assert la is not None and float(la.detach()) == float(lb.detach())
    assert torch.equal(p, q) and not torch.equal(p.detach(), torch.zeros_like(p))
    assert float(a.state[p]["step"]) == 1.0


def test_bench_self_spawns_its_ranks_when_started_without_a_launcher():
    """`python bench.py --gpus 2` (no torchrun around it, no WORLD_SIZE) must start its own two ranks instead of
    dying on an assert. Without a GPU each rank gets as far as the process-group rendezvous (gloo here) and then stops at the
    "needs an AMD GPU" gate — which is what this CPU test looks for; on the GPU box tests/test_hip_dropin.py runs it for real."""
    if torch.cuda.is_available():
        pytest.skip("CPU-only form of the launcher test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert "spawning 2 ranks" in p.stderr, p.stderr[-2000:]
    assert p.stderr.count("bench.py needs an AMD GPU") >= 1, p.stderr[-2000:]          # the ranks came up and met the GPU gate
    assert "but the launcher started" not in p.stderr and "AssertionError" not in p.stderr
    assert p.returncode != 0                                                          # no GPU: the failure is loud, not a fake line


def test_ragged_collate_packs_decoded_images_for_the_gpu_resize():
    """raw="decoded" datasets hand out images at their file size; `ragged_collate` packs a batch as one flat uint8 buffer + an (H, W) table
    (+ labels, weights and any further per-sample items such as the shard-padding flag) — the host half of DeviceResize."""
    from dirhip.datasets import ragged_collate
    rng = np.random.default_rng(0)
    shapes = [(5, 7), (3, 3), (8, 2)]
    imgs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in shapes]
    samples = [(im, np.asarray([20.0 + i], np.float32), np.asarray([1.5], np.float32), i % 2 == 0) for i, im in enumerate(imgs)]
    flat, sizes, labels, weights, valid = ragged_collate(samples)
    assert flat.dtype == torch.uint8 and flat.numel() == sum(h * w * 3 for h, w in shapes) and sizes.tolist() == [list(s) for s in shapes]
    off = 0
    for im in imgs:
        assert torch.equal(flat[off:off + im.numel()].view(im.shape), im)
        off += im.numel()
    assert labels.shape == (3, 1) and weights.shape == (3, 1) and valid.tolist() == [True, False, True]
    assert ragged_collate([s[:3] for s in samples])[3].shape == (3, 1) and len(ragged_collate([s[:3] for s in samples])) == 4


def test_device_resize_refuses_cpu_target():
    from dirhip import _lib as L
    from dirhip.datasets import DeviceResize
    with pytest.raises(L.DirHipError):
        DeviceResize(224, device="cpu")(torch.zeros(12, dtype=torch.uint8), torch.tensor([[2, 2]]))
    lib = L.lib()
    assert lib.dir_resize_ksize(320, 224) == 5 and lib.dir_resize_ksize(100, 224) == 3 and lib.dir_resize_ksize(2048, 224) == 21 and lib.dir_resize_ksize(0, 224) == 0
    assert lib.dir_resize_u8_workspace(256, 224, 5, 1000) > 256 * 2 * 224 * 7 * 4 and lib.dir_resize_u8_workspace(0, 224, 5, 0) == 0
    assert lib.dir_resize_u8(None, 0, None, None, 1, 224, 10, 3, None, 0, None) == -1
    assert lib.dir_sgd_step(None, 1, 0.1, 0.9, 0.0, 0.0, 0, 0, None) == -1


def test_device_image_cache_index_order_and_coverage_logic():
    """datasets.DeviceImageCache (round 5) as a container, on the host: coverage only once every requested sample was stored, one pass = every
    sample of the shard exactly once in a fresh order, labels / weights / the padded-row flag travel with their images, the last batch is ragged."""
    import numpy as np
    import torch
    from dirhip.datasets import DeviceImageCache
    n, s = 23, 4
    imgs = torch.arange(n * s * s * 3, dtype=torch.int64).remainder(251).to(torch.uint8).view(n, s, s, 3)
    labels = torch.arange(n, dtype=torch.float32).view(-1, 1)
    weights = labels * 0.5 + 1
    c = DeviceImageCache(n, s, "cpu")
    first = torch.tensor([4, 0, 22, 9])
    assert not c.covers(first)
    c.put(first, imgs[first], labels[first], weights[first])
    assert c.covers(first) and c.covers([0, 22]) and not c.covers(range(n))
    rest = torch.tensor([i for i in range(n) if i not in set(first.tolist())])
    c.put(rest, imgs[rest], labels[rest], weights[rest])
    assert c.covers(range(n))
    shard = torch.tensor([3, 3, 7, 11, 0, 22, 5, 6, 1, 2, 9])             # (a padded shard repeats samples)
    valid = [True] * 9 + [False, False]
    seen, nb = [], 0
    for x, y, w, v in c.batches(shard, 4, lambda u8: u8.clone(), valid=valid, generator=torch.Generator().manual_seed(0)):
        ids = y.view(-1).long()
        assert torch.equal(x, imgs[ids]) and torch.equal(w, weights[ids]) and v.dtype == torch.bool and len(v) == len(ids) <= 4
        seen += list(zip(ids.tolist(), v.tolist()))
        nb += 1
    assert nb == 3 and sorted(i for i, _ in seen) == sorted(shard.tolist()) and sum(1 for _, ok in seen if not ok) == 2
    order_a = [i for i, _ in seen]
    order_b = [int(i) for b_ in c.batches(shard, 4, lambda u8: u8, valid=valid, generator=torch.Generator().manual_seed(1)) for i in b_[1].view(-1)]
    assert order_a != order_b                                                # a fresh order per pass
    with np.testing.assert_raises(ValueError):
        DeviceImageCache(10 ** 6, 224, "cpu", max_bytes=1 << 30)             # the budget is checked before anything is allocated


def test_shard_subset_carries_the_dataset_index_for_the_cache():
    sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
    from dirhip.train_main import _ShardSubset
    base = [("img%d" % i, float(i), 1.0) for i in range(10)]
    sub = _ShardSubset(base, [7, 2, 2], [True, True, False], with_index=True)
    assert len(sub) == 3 and sub[0] == ("img7", 7.0, 1.0, True, 7) and sub[2] == ("img2", 2.0, 1.0, False, 2)
    assert _ShardSubset(base, [7], [True])[0] == ("img7", 7.0, 1.0, True)
