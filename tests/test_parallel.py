"""world_size-2 (gloo, CPU) tests of the one-process-per-GPU engine: bucketed gradient all-reduce driven by
autograd hooks, parameter broadcast, `.module` / state_dict prefix, zero_grad(set_to_none) interplay,
epoch sharding. The same code runs over RCCL on the GPUs."""
import os

import numpy as np
import torch

import variant_switches as VS  # tools/variant_switches.py: the product package has no setters (conftest puts tools/ on the path)
import torch.nn as nn


def _net():
    torch.manual_seed(3)
    return _net_raw().to(memory_format=torch.channels_last)      # conv weights NHWC-strided like the real model


def _net_raw():
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(),
                         nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 1))


def _worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dirhip.parallel import DataParallelEngine
    net = _net()
    if rank == 1:                                           # rank 1 starts from different weights: must be overwritten
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    eng = DataParallelEngine(net, bucket_mb=0.001)          # tiny buckets -> several collectives per backward
    eng.train()
    opt = torch.optim.SGD(eng.parameters(), lr=0.1)
    d = np.load(os.path.join(tmp, "data.npz"))
    x, y = torch.tensor(d["x"]), torch.tensor(d["y"])
    for it in range(3):
        xs, ys = x[it, rank::world], y[it, rank::world]
        loss = ((eng(xs) - ys) ** 2).mean()
        opt.zero_grad()                                     # set_to_none=True: grads are re-materialised each step
        loss.backward()
        opt.step()
    assert len(eng._buckets) >= 2
    assert eng.stats["bucket_scale_kernels"] == 0           # zero_grad() between forward and backward: not an accumulating pass
    for b in eng._buckets:                                  # gradient views share the parameters' memory layout
        for prm, v in zip(b.params, b.views):
            assert v.stride() == prm.stride() and prm.grad.data_ptr() == v.data_ptr()
    assert all(k.startswith("module.") for k in eng.state_dict())
    torch.save({k: v.clone() for k, v in eng.module.state_dict().items()}, os.path.join(tmp, f"w{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_allreduce_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, (3, 8, 3, 6, 6)).astype(np.float32)
    y = rng.normal(0, 1, (3, 8, 1)).astype(np.float32)
    np.savez(tmp_path / "data.npz", x=x, y=y)
    port = 31000 + int(rng.integers(0, 2000))
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    # single-process reference: the same 3 steps on the full batches (mean over 8 = mean of the two rank means over 4)
    net = _net()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    for it in range(3):
        loss = ((net(torch.tensor(x[it])) - torch.tensor(y[it])) ** 2).mean()
        opt.zero_grad(); loss.backward(); opt.step()
    for (k, a) in net.state_dict().items():
        assert torch.equal(w0[k], w1[k]), f"ranks diverged on {k}"
        assert torch.allclose(w0[k], a, rtol=1e-5, atol=1e-6), k


def test_shard_indices_cover_and_pad():
    from dirhip.parallel import shard_indices
    n, world = 1003, 4
    shards = [shard_indices(n, r, world, epoch_seed=5) for r in range(world)]
    assert len({len(s) for s in shards}) == 1 and len(shards[0]) == 251
    allidx = torch.cat(shards)
    assert set(allidx.tolist()) == set(range(n))
    assert torch.equal(shard_indices(n, 0, 1), torch.arange(n))
    assert not torch.equal(shard_indices(n, 0, world, epoch_seed=5), shard_indices(n, 0, world, epoch_seed=6))


def _accum_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dirhip.parallel import DataParallelEngine
    eng = DataParallelEngine(_net(), bucket_mb=0.001)
    eng.train()
    d = np.load(os.path.join(tmp, "data.npz"))
    x, y = torch.tensor(d["x"]), torch.tensor(d["y"])
    for it in range(2):                                     # two backward passes, NO zero_grad in between
        ((eng(x[it, rank::world]) - y[it, rank::world]) ** 2).mean().backward()
    assert eng.stats["bucket_scale_kernels"] == len(eng._buckets)          # only the second (accumulating) pass scales its buckets
    torch.save({k: p.grad.clone() for k, p in eng.module.named_parameters()}, os.path.join(tmp, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_accumulation_world_size_2_gloo(tmp_path):
    """on the non-RCCL path an accumulating backward pass must not count the already averaged first gradient twice."""
    import torch.multiprocessing as mp
    rng = np.random.default_rng(1)
    x = rng.normal(0, 1, (2, 8, 3, 6, 6)).astype(np.float32)
    y = rng.normal(0, 1, (2, 8, 1)).astype(np.float32)
    np.savez(tmp_path / "data.npz", x=x, y=y)
    port = 33000 + int(rng.integers(0, 2000))
    mp.spawn(_accum_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    net = _net()
    for it in range(2):
        ((net(torch.tensor(x[it])) - torch.tensor(y[it])) ** 2).mean().backward()
    for k, p in net.named_parameters():
        assert torch.equal(g0[k], g1[k]), k
        assert torch.allclose(g0[k], p.grad, rtol=1e-5, atol=1e-6), k


def _ffbb_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dirhip import conv as C
    from dirhip.parallel import DataParallelEngine
    VS.set_wgrad_batched_reduce(True)
    eng = DataParallelEngine(_net(), bucket_mb=0.001)
    # live collectives start from per-parameter hooks, before the end of the backward pass: the one-launch-per-pass weight-gradient
    # reduction (conv.set_wgrad_batched_reduce) must be off under them
    assert C._WGRAD_BATCH["on"] is False
    # ... and stays ineffective if somebody turns it back on while the engine lives (guarded at the point of use)
    VS.set_wgrad_batched_reduce(True)
    assert C._COLLECTIVES_LIVE[0] > 0 and C._wgrad_batch_slot(torch.empty(4), True, 16) is None
    eng.train()
    d = np.load(os.path.join(tmp, "data.npz"))
    x, y = torch.tensor(d["x"]), torch.tensor(d["y"])
    losses = [((eng(x[it, rank::world]) - y[it, rank::world]) ** 2).mean() for it in range(2)]       # forward, forward ...
    for l in losses:                                                                                  # ... backward, backward
        l.backward()
    # the first pass is a plain one (pre-scaled output gradient), the second finds the averaged gradients held and takes the accumulation form
    assert eng.stats["bucket_scale_kernels"] == len(eng._buckets)
    torch.save({k: p.grad.clone() for k, p in eng.module.named_parameters()}, os.path.join(tmp, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_two_forwards_then_two_backwards_world_size_2_gloo(tmp_path):
    """the accumulation decision is taken per BACKWARD pass (reset when a pass finishes), so forward, forward, backward, backward
    ends with the sum of the two rank-averaged gradients — not world * mean_1 + mean_2."""
    import torch.multiprocessing as mp
    rng = np.random.default_rng(2)
    x = rng.normal(0, 1, (2, 8, 3, 6, 6)).astype(np.float32)
    y = rng.normal(0, 1, (2, 8, 1)).astype(np.float32)
    np.savez(tmp_path / "data.npz", x=x, y=y)
    port = 35000 + int(rng.integers(0, 2000))
    mp.spawn(_ffbb_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    net = _net()
    for it in range(2):
        ((net(torch.tensor(x[it])) - torch.tensor(y[it])) ** 2).mean().backward()
    for k, p in net.named_parameters():
        assert torch.equal(g0[k], g1[k]), k
        assert torch.allclose(g0[k], p.grad, rtol=1e-5, atol=1e-6), k


class _NetWithUnused(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.body = _net_raw()
        self.unused = nn.Linear(4, 4)                                 # never part of the graph: its gradients must arrive as zeros

    def forward(self, x):
        return self.body(x)


def _edge_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dirhip.parallel import DataParallelEngine
    eng = DataParallelEngine(_NetWithUnused(), bucket_mb=0.001)
    eng.train()
    opt = torch.optim.SGD(eng.parameters(), lr=0.05)
    d = np.load(os.path.join(tmp, "data.npz"))
    x, y = torch.tensor(d["x"]), torch.tensor(d["y"])
    for it in range(3):
        loss = ((eng(x[it, rank::world]) - y[it, rank::world]) ** 2).mean()
        opt.zero_grad(set_to_none=False)                              # gradients stay allocated (zero-valued): classed as accumulating, same result
        loss.backward()
        opt.step()
    un = eng.module.unused
    assert un.weight.grad is not None and float(un.weight.grad.abs().sum()) == 0.0 and float(un.bias.grad.abs().sum()) == 0.0
    torch.save({k: v.clone() for k, v in eng.module.state_dict().items()}, os.path.join(tmp, f"w{rank}.pt"))
    dist.destroy_process_group()


def test_unused_parameters_and_kept_zero_gradients_world_size_2_gloo(tmp_path):
    """Parameters outside the graph get zero gradients in their bucket (every bucket's collective still runs on every rank: no hang), and a
    loop that keeps its gradients allocated (zero_grad(set_to_none=False)) ends where the set-to-none loop and the single process end."""
    import torch.multiprocessing as mp
    rng = np.random.default_rng(4)
    x = rng.normal(0, 1, (3, 8, 3, 6, 6)).astype(np.float32)
    y = rng.normal(0, 1, (3, 8, 1)).astype(np.float32)
    np.savez(tmp_path / "data.npz", x=x, y=y)
    port = 37000 + int(rng.integers(0, 2000))
    mp.spawn(_edge_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    net = _NetWithUnused()
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    for it in range(3):
        loss = ((net(torch.tensor(x[it])) - torch.tensor(y[it])) ** 2).mean()
        opt.zero_grad(); loss.backward(); opt.step()
    for k, a in net.state_dict().items():
        assert torch.equal(w0[k], w1[k]), f"ranks diverged on {k}"
        assert torch.allclose(w0[k], a, rtol=1e-5, atol=1e-6), k
