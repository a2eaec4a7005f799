"""-m gpu: the all-taps 3x3 weight gradient (dir_conv_wgrad3x3: LDS-DMA staging in natural layout + transposing LDS reads)
against the per-tap kernel it replaces (dir_conv_wgrad) and against a float64 reference; plus the probe that pins the lane
mapping of ds_read_b64_tr_b16 the kernel is built on.

Replaces the weight-gradient half of nn.Conv2d's autograd for conv2 of the Bottlenecks (imdb-wiki-dir/resnet.py:46-47)."""
import numpy as np
import pytest
import torch

import variant_switches as VS  # tools/variant_switches.py: the product package has no setters (conftest puts tools/ on the path)

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def test_transposing_lds_read_lane_mapping():
    """Within each 16-lane group, lane i supplies the address of 4 consecutive 16-bit elements (matrix row i >> 2, columns
    4 (i & 3) .. + 3) and receives column i of the 4 x 16 block, rows 0 .. 3 — with arbitrary (8-byte aligned) row addresses."""
    import os
    import sys
    from conftest import ROOT
    from dirhip import _lib as L
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import toolslib                                    # the probe lives in tools/lib/libdir_hip_tools.so, not in the product library
    rng = np.random.default_rng(0)
    for trial in range(4):
        if trial == 0:       # the contiguous 4 x 16 block per group of the guide: lds[(l & 15) + 16 j + 64 (l >> 4)]
            addr = np.array([2 * (64 * (l >> 4) + 16 * ((l & 15) >> 2) + 4 * (l & 3)) for l in range(64)], np.int32)
        else:                # scattered rows
            addr = (rng.integers(0, 2040, 64) * 8).astype(np.int32)
        out = torch.empty(64 * 4, dtype=torch.int16, device="cuda")
        a = torch.as_tensor(addr).cuda()
        L.check(toolslib.lib().dir_probe_tr16(L.ptr(a), L.ptr(out), L.stream_ptr(a.device)), "dir_probe_tr16")
        got = out.cpu().numpy().astype(np.int64).reshape(64, 4) & 0xffff
        want = np.zeros((64, 4), np.int64)
        for l in range(64):
            g, i = l >> 4, l & 15
            for j in range(4):
                src = 16 * g + 4 * j + (i >> 2)            # the lane that addressed row j's column quad holding column i
                want[l, j] = addr[src] // 2 + (i & 3)
        assert np.array_equal(got, want), (trial, got[:20], want[:20])
        if trial == 0:
            l = np.arange(64)[:, None]
            j = np.arange(4)[None, :]
            assert np.array_equal(got, (l & 15) + 16 * j + 64 * (l >> 4))


SHAPES = [  # (N, C_in, C_out, H)
    (8, 64, 64, 56), (5, 128, 128, 28), (6, 256, 256, 14), (7, 512, 512, 7), (3, 64, 128, 28), (4, 192, 64, 14), (1, 64, 64, 7),
    (33, 128, 64, 7),
]


@pytest.mark.parametrize("n,cin,cout,hw", SHAPES)
def test_all_taps_wgrad_vs_per_tap_kernel_and_float64(n, cin, cout, hw):
    from dirhip import conv as C
    g = torch.Generator(device="cuda").manual_seed(n + cin + cout + hw)
    x = _cl((torch.randn(n, cin, hw, hw, device="cuda", generator=g)).to(torch.bfloat16))
    dy = _cl((torch.randn(n, cout, hw, hw, device="cuda", generator=g) * 0.5).to(torch.bfloat16))
    # a transpose-detecting pattern on top of the noise: channel- and position-dependent ramps
    x = _cl((x.float() + torch.arange(cin, device="cuda").view(1, -1, 1, 1) * 0.01
             + torch.arange(hw, device="cuda").view(1, 1, -1, 1) * 0.02 - torch.arange(hw, device="cuda").view(1, 1, 1, -1) * 0.03).to(torch.bfloat16))
    prev = VS.set_wgrad3_all_taps(True)
    try:
        dw = C.conv2d_wgrad(dy, x, 3, 1, 1)
        VS.set_wgrad3_all_taps(False)
        dw_tap = C.conv2d_wgrad(dy, x, 3, 1, 1)
    finally:
        VS.set_wgrad3_all_taps(prev)
    assert dw.shape == (cout, cin, 3, 3) and dw.is_contiguous(memory_format=torch.channels_last)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), stride=1, padding=1)
    scale = float(ref.abs().max())
    e_new = float((dw.double() - ref).abs().max()) / scale
    e_tap = float((dw_tap.double() - ref).abs().max()) / scale
    # float32 accumulation of N*H*W bf16 products in two different orders: both within 1e-5 of the float64 result's scale
    assert e_new <= 1e-5 and e_tap <= 1e-5, (e_new, e_tap)
    assert_close(dw.cpu().numpy(), dw_tap.cpu().numpy(), rtol=1e-4, atol_scale=1e-5, msg="all-taps vs per-tap")


def test_all_taps_wgrad_is_deterministic_and_ignores_poisoned_workspace():
    from dirhip import _lib as L
    n, c, hw = 16, 128, 28
    g = torch.Generator(device="cuda").manual_seed(9)
    x = _cl(torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16))
    dy = _cl(torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16))
    nbytes = L.lib().dir_conv_wgrad3x3_workspace(n, hw, hw, c, c)
    assert nbytes > 0
    outs = []
    for fill in (0, 0x7f):
        ws = torch.full((nbytes,), fill, dtype=torch.uint8, device="cuda")          # 0x7f7f7f7f = a huge float: every slot must be overwritten
        dw = torch.empty((c, c, 3, 3), dtype=torch.float32, device="cuda").contiguous(memory_format=torch.channels_last)
        L.check(L.lib().dir_conv_wgrad3x3(L.ptr(dy), L.ptr(x), L.ptr(dw), n, hw, hw, c, c, L.ptr(ws), nbytes, L.stream_ptr(x.device)), "wgrad3x3")
        outs.append(dw.clone())
    assert torch.equal(outs[0], outs[1])
    assert L.lib().dir_conv_wgrad3x3_workspace(n, hw, hw + 1, c, c) == 0 and L.lib().dir_conv_wgrad3x3_workspace(n, 20, 20, c, c) == 0
    assert L.lib().dir_conv_wgrad3x3_workspace(n, hw, hw, 96, c) == 0


@pytest.mark.parametrize("c,hw", [(64, 56), (128, 28), (256, 14), (512, 7)])
def test_all_taps_wgrad_full_size_vs_per_tap_and_linearity(c, hw):
    """BASELINE batch (256): the two kernels agree, and the result is linear in dY (dW(a + b) = dW(a) + dW(b) up to float32 sums)."""
    from dirhip import conv as C
    n = 256
    g = torch.Generator(device="cuda").manual_seed(c + hw)
    x = _cl(torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16))
    a = _cl((torch.randn(n, c, hw, hw, device="cuda", generator=g) * 0.25).to(torch.bfloat16))
    b = _cl((torch.randn(n, c, hw, hw, device="cuda", generator=g) * 0.25).to(torch.bfloat16))
    ab = _cl((a.float() + b.float()).to(torch.bfloat16))
    exact = torch.equal(ab.float(), a.float() + b.float())
    dwa, dwb, dwab = C.conv2d_wgrad(a, x, 3, 1, 1), C.conv2d_wgrad(b, x, 3, 1, 1), C.conv2d_wgrad(ab, x, 3, 1, 1)
    prev = VS.set_wgrad3_all_taps(False)
    try:
        dwa_tap = C.conv2d_wgrad(a, x, 3, 1, 1)
    finally:
        VS.set_wgrad3_all_taps(prev)
    scale = float(dwa.abs().max())
    assert float((dwa - dwa_tap).abs().max()) <= 2e-5 * scale
    if exact:
        assert float((dwab - (dwa + dwb)).abs().max()) <= 2e-5 * float(dwab.abs().max())
    else:                                                        # a + b rounded to bf16: compare against the rounded operand's own sum
        ref = C.conv2d_wgrad(_cl((ab.float() - a.float()).to(torch.bfloat16)), x, 3, 1, 1) + dwa
        assert float((dwab - ref).abs().max()) <= 1e-3 * float(dwab.abs().max())
