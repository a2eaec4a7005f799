"""-m gpu: the patch-staged 3x3 / stride-1 convolution kernel (conv3x3_patch_kernel: input patch of a chunk of image rows staged
in LDS once per 64-channel block, nine taps as row offsets) against the per-tap implicit-GEMM kernels it replaces on the
56^2 / 28^2 / 14^2 maps, and against a float32 reference. Forward of conv2 of the Bottlenecks and, with rotated weights, its
data gradient (imdb-wiki-dir/resnet.py:46-47)."""
import numpy as np
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _run(x, w, variant, want_stats=True):
    from dirhip import _lib as L
    n, cin, h, _ = x.shape
    cout = w.shape[0]
    y = torch.empty((n, cout, h, h), dtype=torch.bfloat16, device=x.device).contiguous(memory_format=torch.channels_last)
    used = L.lib().dir_conv_plan_rows(n, h, h, cin, cout, 3, 3, 1, 1, 0, variant)
    assert used > 0
    st = torch.full((used, 2, cout), float("nan"), dtype=torch.float32, device=x.device) if want_stats else None
    L.check(L.lib().dir_conv_fwd_variant(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), used if want_stats else 0, n, h, h, cin, cout, 3, 3, 1, 1, variant,
                                         L.stream_ptr(x.device)), "dir_conv_fwd_variant")
    return y, (st[:used] if want_stats else None)


@pytest.mark.parametrize("n,cin,cout,hw", [(4, 64, 64, 56), (3, 128, 128, 28), (5, 256, 256, 14), (2, 64, 128, 28), (3, 192, 64, 14),
                                           (1, 64, 64, 14), (2, 128, 256, 56)])
def test_patch_kernel_vs_per_tap_kernel_and_float32(n, cin, cout, hw):
    g = torch.Generator(device="cuda").manual_seed(n + cin + cout + hw)
    x = _cl(torch.randn(n, cin, hw, hw, device="cuda", generator=g).to(torch.bfloat16))
    # transpose-detecting weights: every (co, ci, r, s) different in a structured way on top of noise
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / np.sqrt(cin * 9)
    w = w + (torch.arange(9, device="cuda").view(1, 1, 3, 3) - 4) * 0.01 + torch.arange(cin, device="cuda").view(1, -1, 1, 1) * 1e-4
    w = _cl(w.to(torch.bfloat16))
    y3, s3 = _run(x, w, 3)
    y1, s1 = _run(x, w, 1)
    y0, s0 = _run(x, w, 0)
    assert torch.equal(y0, y3) and torch.equal(s0, s3)               # the heuristic takes the patch kernel for these shapes
    ref = torch.nn.functional.conv2d(x.float(), w.float(), padding=1)
    assert_close(y3.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol_scale=4e-3, msg="vs float32")
    if cin == 64:
        assert torch.equal(y3, y1)                                   # one channel block: identical accumulation order, bit for bit
    else:
        d = (y3.float() - y1.float()).abs()
        assert float((d > 0).float().mean()) < 0.05                  # float32 sums in another K order: a few bf16 roundings differ
        assert_close(y3.float().cpu().numpy(), y1.float().cpu().numpy(), rtol=8e-3, atol_scale=1e-3, msg="vs per-tap kernel")
    # BatchNorm statistics of the rounded outputs: per-chunk partials sum to the column sums
    yf = y3.float()
    assert not torch.isnan(s3).any()
    tot = s3.double().sum(0)
    assert_close(tot[0].cpu().numpy(), yf.double().sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sum")
    assert_close(tot[1].cpu().numpy(), (yf.double() ** 2).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sumsq")


def test_patch_kernel_fused_data_gradient_epilogue():
    """The data-gradient use: rotated weights, + shortcut gradient, ReLU mask, and the BatchNorm-backward sums, through conv2d_igemm."""
    from dirhip.conv import conv2d_igemm
    from dirhip.bn import BwdLink
    n, c, hw = 3, 128, 28
    g = torch.Generator(device="cuda").manual_seed(5)
    dy = _cl(torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16))
    w = _cl((torch.randn(c, c, 3, 3, device="cuda", generator=g) / np.sqrt(c * 9)).to(torch.bfloat16))
    addend = _cl(torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16))
    mask = _cl(torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16))
    bnx = _cl(torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16))
    link = BwdLink()
    link.x = bnx
    link.recompute_mask = False
    y = conv2d_igemm(dy, w, 1, 1, addend=addend, relu_mask=mask, bn_link=link)
    ref = torch.nn.functional.conv2d(dy.float(), w.float(), padding=1).to(torch.bfloat16).float() + addend.float()
    ref = torch.where(mask.float() > 0, ref.to(torch.bfloat16).float(), torch.zeros_like(ref))
    assert_close(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol_scale=4e-3, msg="fused dgrad")
    got = link.partial.double().sum(0)
    assert_close(got[0].cpu().numpy(), y.double().sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sum g")
    assert_close(got[1].cpu().numpy(), (y.double() * bnx.double()).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sum g*x")


@pytest.mark.parametrize("c,hw", [(64, 56), (128, 28), (256, 14)])
def test_patch_kernel_full_size_vs_per_tap(c, hw):
    """BASELINE batch (256): every output chunk / tile of the patch-staged kernel against the per-tap kernel, and the partial
    statistics lists of the two tilings sum to the same column sums."""
    n = 256
    g = torch.Generator(device="cuda").manual_seed(c * hw)
    x = _cl(torch.randn(n, c, hw, hw, device="cuda", generator=g).to(torch.bfloat16))
    w = _cl((torch.randn(c, c, 3, 3, device="cuda", generator=g) / np.sqrt(c * 9)).to(torch.bfloat16))
    y3, s3 = _run(x, w, 3)
    y1, s1 = _run(x, w, 1)
    if c == 64:
        assert torch.equal(y3, y1)
    else:
        d = (y3.float() - y1.float()).abs()
        assert float((d > 0).float().mean()) < 0.05 and float(d.max()) <= 0.02 * float(y1.float().abs().max())
    t3, t1 = s3.double().sum(0), s1.double().sum(0)
    assert_close(t3.cpu().numpy(), t1.cpu().numpy(), rtol=2e-4, atol_scale=2e-4, msg="statistics")
