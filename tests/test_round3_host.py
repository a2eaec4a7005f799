"""CPU tests of round-3 host logic: the bf16-rounding emulation used as parity oracle (oracle/bf16_emul.py), the gradient-sink
registry of the data-parallel engine (dirhip/gradsink.py), the persistent ring kernel's tile assignment restated on the CPU, and the
committed round-3 goldens' internal consistency."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN  # noqa: F401


def test_round_bf16_is_nearest_even_both_ways():
    from oracle.bf16_emul import RoundBF16
    # 1 + 2^-8 is exactly between the bf16 neighbours 1 and 1 + 2^-7: ties to even -> 1; 1 + 3 * 2^-8 -> 1 + 2^-6 (even mantissa)
    x = torch.tensor([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -(1.0 + 2.0 ** -8), 1.0 + 2.0 ** -8 + 2.0 ** -20, 3.0e38, 0.0], dtype=torch.float64, requires_grad=True)
    y = RoundBF16.apply(x)
    assert y.dtype == torch.float64
    assert y.tolist()[:4] == [1.0, 1.0 + 2.0 ** -6, -1.0, 1.0 + 2.0 ** -7]
    y.backward(torch.tensor([1.0 + 2.0 ** -8, 0.3, 1.0, 1.0, 1.0, 1.0], dtype=torch.float64))
    assert x.grad[0].item() == 1.0 and x.grad[1].item() == float(torch.tensor(0.3).bfloat16())


def _block_pair(make_ref_block):
    from oracle.bf16_emul import bf16_points
    from oracle.torch_oracle import _Bottleneck
    torch.manual_seed(3)
    ds = torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, 2, bias=False), torch.nn.BatchNorm2d(128))
    ob = _Bottleneck(64, 32, 2, ds).double()
    with torch.no_grad():
        for m in ob.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    rb = make_ref_block(ob)
    hooks = bf16_points(ob)
    assert len(hooks) == 4 + 2 + 1                      # four convolutions, bn1 / bn2, the block output
    bf16_points(rb)
    return ob, rb


def test_emulated_block_rounds_where_the_product_stores():
    """Weights are rounded in place; conv outputs, relu(bn1), relu(bn2) and the block output are bf16-representable; bn3 and the
    downsample BatchNorm are NOT rounded on their own (the join adds them in float32)."""
    ob, _ = _block_pair(lambda ob: ob)
    seen = {}
    for name, m in ob.named_modules():
        if name:
            m.register_forward_hook(lambda mod, i, o, name=name: seen.__setitem__(name, o.detach()))
    x = torch.randn(2, 64, 8, 8, dtype=torch.float64).float().bfloat16().double()
    y = ob.train()(x)

    def is_bf16(t):
        return torch.equal(t, t.float().bfloat16().double())
    assert all(is_bf16(m.weight) for m in ob.modules() if isinstance(m, torch.nn.Conv2d))
    for k in ("conv1", "conv2", "conv3", "downsample.0", "bn1", "bn2"):
        assert is_bf16(seen[k]), k
    assert not is_bf16(seen["bn3"]) and not is_bf16(seen["downsample.1"])
    assert is_bf16(y.detach())


@pytest.mark.needs_reference
def test_emulation_on_the_port_equals_emulation_on_the_live_reference():
    """The GPU box has no /root/reference: the teacher-forced bf16 test runs the emulation on oracle.torch_oracle's port of the
    block. Here the same emulation on the reference's OWN Bottleneck (same weights, same input): identical float64 outputs and
    input gradients."""
    from oracle import refshim
    ref = refshim.load("imdb-wiki-dir")

    def make(ob):
        ds = torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, 2, bias=False), torch.nn.BatchNorm2d(128))
        rb = ref.resnet.Bottleneck(64, 32, 2, ds).double()
        rb.load_state_dict(ob.state_dict())
        return rb
    ob, rb = _block_pair(make)
    x = torch.randn(3, 64, 8, 8, dtype=torch.float64).float().bfloat16().double()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = ob.train()(xa), rb.train()(xb)
    assert torch.equal(ya, yb)
    g = torch.randn_like(ya)
    ya.backward(g); yb.backward(g)
    assert torch.equal(xa.grad, xb.grad)


def test_gradsink_registry_hands_out_fresh_bucket_views():
    import sys, os
    from conftest import PKG
    sys.path.insert(0, PKG)
    from dirhip import gradsink
    flat = torch.zeros(64 + 64 * 3)
    p = torch.nn.Parameter(torch.randn(4, 3, 2, 2).contiguous(memory_format=torch.channels_last))
    q = torch.nn.Parameter(torch.randn(7))
    gradsink.register(p, lambda: flat[:48].as_strided(p.shape, p.stride()))
    try:
        a = gradsink.out_for(p, (4, 3, 2, 2), torch.device("cpu"), torch.channels_last)
        b = gradsink.out_for(p, (4, 3, 2, 2), torch.device("cpu"), torch.channels_last)
        assert a is not b and a.data_ptr() == b.data_ptr() == flat.data_ptr() and a.stride() == p.stride()
        a.fill_(2.0)
        assert float(flat[:48].sum()) == 96.0 and float(flat[48:].abs().sum()) == 0.0
        assert gradsink.out_for(q, (7,), torch.device("cpu")).data_ptr() != flat.data_ptr()          # unregistered: a new tensor
        assert gradsink.out_for(p, (5, 3, 2, 2), torch.device("cpu")).numel() == 60                    # geometry mismatch: a new tensor
        f = gradsink.lookup(p)
        assert gradsink.out_for(f, (48,), torch.device("cpu")).numel() == 48                           # (non-contiguous slot: falls back)
    finally:
        gradsink.unregister([p])
    assert gradsink.lookup(p) is None


@pytest.mark.parametrize("nblocks", [1, 7, 8, 9, 98, 392, 784, 1568, 3136, 12544, 255, 1001])
def test_conv_workgroup_remap_is_a_permutation(nblocks):
    """csrc/dir_conv.hip (every tile kernel, `lin`): workgroup b runs on XCD b % 8 (observed placement), so the linear tile index is
    remapped so that each XCD walks ONE contiguous chunk of tiles (the N tiles of an M tile share that XCD's L2). Restated here from the
    kernels' expression: it must be a permutation of 0 .. nblocks - 1 with contiguous per-XCD chunks whose sizes differ by at most one."""
    q, r = nblocks // 8, nblocks % 8
    lin = np.empty(nblocks, np.int64)
    for b in range(nblocks):
        xcd, i = b % 8, b // 8
        lin[b] = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + i
    assert np.array_equal(np.sort(lin), np.arange(nblocks))
    for xcd in range(8):
        mine = np.sort(lin[xcd::8])
        assert len(mine) in (q, q + 1) and (len(mine) == 0 or np.array_equal(mine, np.arange(mine[0], mine[0] + len(mine))))


def test_round3_goldens_are_consistent(golden):
    e, r = golden("step0_b64_bf16emul.npz"), golden("step0_b64.npz")
    assert np.array_equal(e["in_labels"], r["in_labels"]) and list(e["param_names"]) == list(r["param_names"])
    # the emulated run is a bf16 evaluation of the same step: ~1e-4 from the float32 loss, ~10 % in the encoding — and its float32-
    # arithmetic twin sits at the same distance from it as any other evaluation of the bf16 graph (the noise floor the GPU test uses)
    assert 1e-5 < abs(float(e["emul_loss"]) - float(r["ref_loss"])) / float(r["ref_loss"]) < 1e-3
    d = np.linalg.norm(e["emul32_encoding"].astype(np.float64) - e["emul_encoding"]) / np.linalg.norm(e["emul_encoding"])
    assert 1e-2 < d < 0.15
    b = golden("step0_b256.npz")
    assert b["ref_pred"].shape == (256, 1) and abs(float(b["ref_loss"]) - float(b["ref64_loss"])) / float(b["ref64_loss"]) < 1e-6
    m = golden("multistep_b32.npz")
    assert m["ref_losses"].shape == (4,) and abs(m["ref_losses"][0] - m["ref64_losses"][0]) / m["ref64_losses"][0] < 1e-6
