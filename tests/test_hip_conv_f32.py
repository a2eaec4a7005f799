"""-m gpu: the exact-float32 MFMA convolutions / pools of the parity mode (dir_conv_f32_*, dir_*pool*_f32_*) against
float64 torch-CPU references of the same ops: forward, data gradient and weight gradient of every layer family of the
reference's ResNet (7x7/2 stem with Cin = 3, 1x1, 3x3, stride 1 / 2) plus ragged / odd-sized / non-multiple-of-64 shapes.
Tolerance: float32 accumulation noise only (1e-5 of the element + 2e-6 of the array's scale)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu

CASES = [  # (N, Cin, H, W, Cout, k, stride, pad)
    (2, 3, 32, 40, 64, 7, 2, 3),          # stem geometry
    (2, 64, 14, 14, 64, 1, 1, 0), (2, 64, 15, 13, 128, 1, 2, 0), (3, 64, 14, 14, 64, 3, 2, 1),
    (2, 32, 9, 11, 48, 3, 1, 1),          # channel counts the bf16 kernel does not take
    (1, 128, 7, 7, 256, 3, 1, 1), (2, 64, 8, 8, 64, 3, 2, 1), (1, 256, 7, 5, 64, 1, 1, 0), (5, 16, 5, 3, 8, 3, 2, 1),
    (1, 64, 1, 1, 64, 1, 1, 0),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_f32_fwd_dgrad_wgrad_vs_float64(case):
    from dirhip.conv_f32 import conv2d_f32_dgrad, conv2d_f32_fwd, conv2d_f32_wgrad
    n, cin, h, w, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) * 7 + 1)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5         # asymmetric in every index
    xd = x.double().requires_grad_(True)
    wd = wt.double().requires_grad_(True)
    ref = F.conv2d(xd, wd, None, stride, pad)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    dev = "cuda"
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    wg = wt.to(dev).contiguous(memory_format=torch.channels_last)
    dyg = dy.to(dev).contiguous(memory_format=torch.channels_last)
    y = conv2d_f32_fwd(xg, wg, stride, pad)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert_close(y.cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol_scale=2e-6, msg=f"fwd {case}")
    dx = conv2d_f32_dgrad(dyg, wg, (h, w), stride, pad)
    assert_close(dx.cpu().numpy(), xd.grad.numpy(), rtol=1e-5, atol_scale=2e-6, msg=f"dgrad {case}")
    dw = conv2d_f32_wgrad(dyg, xg, (k, k), stride, pad)
    assert dw.shape == wt.shape
    assert_close(dw.cpu().numpy(), wd.grad.numpy(), rtol=1e-5, atol_scale=2e-6, msg=f"wgrad {case}")


TILE_CASES = [  # (N, Cin, H, W, Cout, k, stride, pad): geometries the tile kernels take (whole 16-channel K-steps)
    (2, 64, 14, 14, 64, 1, 1, 0), (2, 64, 15, 13, 128, 1, 2, 0), (3, 64, 14, 14, 64, 3, 2, 1), (2, 32, 9, 11, 48, 3, 1, 1),
    (1, 128, 7, 7, 256, 3, 1, 1), (2, 64, 8, 8, 64, 3, 2, 1), (1, 256, 7, 5, 64, 1, 1, 0), (5, 16, 5, 3, 16, 3, 2, 1),
    (3, 256, 14, 14, 1024, 1, 1, 0), (2, 512, 7, 7, 512, 3, 1, 1), (2, 128, 28, 28, 128, 3, 2, 1), (1, 64, 1, 1, 64, 1, 1, 0),
    (2, 48, 10, 6, 80, 3, 1, 1),            # ragged tiles in every direction (Cout = 80, M = 120)
    (2, 32, 12, 10, 32, 1, 2, 0),           # 1x1 stride 2: three of the four parity classes of the data gradient have no tap (zeros)
    (2, 16, 9, 7, 32, 3, 2, 1),             # stride 2 on an odd-sized map: the masked (not the parity-class) data gradient
]


@pytest.mark.parametrize("case", TILE_CASES)
def test_conv_f32_tile_kernels_vs_gather_and_float64(case):
    """Round 6: the 128 x 128 tile kernels (LDS-DMA staging, two stages) against (a) the element-gather kernels — forward and data
    gradient BIT-IDENTICAL (same k order on the same MFMA), also with the fused store epilogue (addend, compact stride-2 addend, ReLU
    mask) — and (b) float64 torch-CPU references of all three GEMMs at the float32 accumulation-noise tolerance."""
    from dirhip.conv_f32 import GATHER, TILE, conv2d_f32_dgrad, conv2d_f32_fwd, conv2d_f32_wgrad
    n, cin, h, w, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) * 11 + 3)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    ref = F.conv2d(xd, wd, None, stride, pad)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)     # noqa: E731
    xg, wg, dyg = cl(x), cl(wt), cl(dy)
    y_t, y_g = conv2d_f32_fwd(xg, wg, stride, pad, variant=TILE), conv2d_f32_fwd(xg, wg, stride, pad, variant=GATHER)
    assert torch.equal(y_t, y_g), f"fwd tile != gather {case}: {(y_t - y_g).abs().max().item()}"
    assert torch.equal(conv2d_f32_fwd(xg, wg, stride, pad), y_t)              # the product's choice is the tile kernel here
    assert_close(y_t.cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol_scale=2e-6, msg=f"fwd {case}")
    dx_t = conv2d_f32_dgrad(dyg, wg, (h, w), stride, pad, variant=TILE)
    dx_g = conv2d_f32_dgrad(dyg, wg, (h, w), stride, pad, variant=GATHER)
    assert torch.equal(dx_t, dx_g), f"dgrad tile != gather {case}: {(dx_t - dx_g).abs().max().item()}"
    assert_close(dx_t.cpu().numpy(), xd.grad.numpy(), rtol=1e-5, atol_scale=2e-6, msg=f"dgrad {case}")
    # fused store epilogue
    add = cl(torch.randn(n, cin, h, w, generator=g))
    msk = cl(torch.randn(n, cin, h, w, generator=g))
    kw = dict(addend=add, relu_mask=msk)
    if h % 2 == 0 and w % 2 == 0:
        kw["addend_s2"] = cl(torch.randn(n, cin, h // 2, w // 2, generator=g))
    f_t = conv2d_f32_dgrad(dyg, wg, (h, w), stride, pad, variant=TILE, **kw)
    f_g = conv2d_f32_dgrad(dyg, wg, (h, w), stride, pad, variant=GATHER, **kw)
    assert torch.equal(f_t, f_g), f"fused dgrad tile != gather {case}"
    exp = dx_t + add
    if "addend_s2" in kw:
        exp[:, :, ::2, ::2] += kw["addend_s2"]
    assert torch.equal(f_t, torch.where(msk > 0, exp, torch.zeros_like(exp)))
    dw_t = conv2d_f32_wgrad(dyg, xg, (k, k), stride, pad, variant=TILE)
    assert_close(dw_t.cpu().numpy(), wd.grad.numpy(), rtol=1e-5, atol_scale=2e-6, msg=f"wgrad {case}")
    assert torch.equal(conv2d_f32_wgrad(dyg, xg, (k, k), stride, pad, variant=TILE), dw_t)       # fixed-order split-K: reproducible


@pytest.mark.parametrize("case", [(2, 64, 14, 14, 64, 1, 1, 0), (3, 64, 14, 14, 128, 3, 2, 1), (2, 48, 10, 6, 80, 3, 1, 1), (4, 256, 14, 14, 1024, 1, 1, 0)])
def test_conv_f32_forward_carries_the_batchnorm_statistics(case):
    """dir_conv_f32_fwd_stats: y identical to the plain forward, and the per-slab (sum, sum of squares) partials add up to the column sums of y
    (float32 sums over 64 rows, compared in float64 at 1e-6 of the column's scale); rows past M contribute nothing."""
    from dirhip.conv_f32 import conv2d_f32_fwd, stats_fusable
    n, cin, h, w, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.1).cuda().contiguous(memory_format=torch.channels_last)
    assert stats_fusable(x, wt)
    y, part = conv2d_f32_fwd(x, wt, stride, pad, want_stats=True)
    assert torch.equal(y, conv2d_f32_fwd(x, wt, stride, pad))
    m = y.numel() // cout
    assert part.shape == ((m + 127) // 128 * 2, 2, cout)
    y2 = y.permute(0, 2, 3, 1).reshape(m, cout).double()
    s1, s2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    scale = y2.abs().sum(0)
    assert ((s1 - y2.sum(0)).abs() <= 1e-6 * scale + 1e-12).all()
    assert ((s2 - (y2 * y2).sum(0)).abs() <= 1e-6 * (y2 * y2).sum(0) + 1e-12).all()
    # per slab: exactly the rows of that slab
    for r in range(part.shape[0]):
        rows = y2[r * 64:(r + 1) * 64]
        assert ((part[r, 0].double() - rows.sum(0)).abs() <= 2e-6 * rows.abs().sum(0) + 1e-12).all(), r
    # the stem (Cin = 3) runs on the gather kernel: no fused statistics, the BatchNorm counts itself
    xs = torch.randn(2, 3, 16, 16).cuda().contiguous(memory_format=torch.channels_last)
    ws = torch.randn(64, 3, 7, 7).cuda().contiguous(memory_format=torch.channels_last)
    assert conv2d_f32_fwd(xs, ws, 2, 3, want_stats=True)[1] is None


def test_conv_f32_tile_variant_refuses_what_it_does_not_take():
    from dirhip import _lib as L
    from dirhip.conv_f32 import TILE, conv2d_f32_fwd
    x = torch.randn(2, 3, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)          # the stem: Cin = 3
    w = torch.randn(64, 3, 7, 7, device="cuda").contiguous(memory_format=torch.channels_last)
    with pytest.raises(L.DirHipError):
        conv2d_f32_fwd(x, w, 2, 3, variant=TILE)
    assert conv2d_f32_fwd(x, w, 2, 3).shape == (2, 64, 8, 8)                                            # auto: the gather kernel


def test_conv_f32_autograd_node_and_determinism():
    """conv_f32 (the autograd node resnet.py uses in parity mode) == F.conv2d in float64, also for a bf16 caller (casts
    around the float32 kernels), and two runs are bit-identical (fixed-order split-K, no atomics)."""
    import torch.nn as nn
    from dirhip.conv_f32 import conv_f32
    torch.manual_seed(4)
    conv = nn.Conv2d(64, 128, 3, 2, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(4, 64, 28, 28, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dy = torch.randn(4, 128, 14, 14, device="cuda")
    outs = []
    for _ in range(2):
        conv.weight.grad = None
        x.grad = None
        y = conv_f32(x, conv)
        y.backward(dy)
        outs.append((y.detach().clone(), x.grad.clone(), conv.weight.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    xd = x.detach().double().cpu().requires_grad_(True)
    wd = conv.weight.detach().double().cpu().requires_grad_(True)
    ref = F.conv2d(xd, wd, None, 2, 1)
    ref.backward(dy.double().cpu())
    assert_close(outs[0][0].cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol_scale=2e-6, msg="y")
    assert_close(outs[0][1].cpu().numpy(), xd.grad.numpy(), rtol=1e-5, atol_scale=2e-6, msg="dx")
    assert_close(outs[0][2].cpu().numpy(), wd.grad.numpy(), rtol=1e-5, atol_scale=2e-6, msg="dw")
    xb = x.detach().to(torch.bfloat16).requires_grad_(True)
    yb = conv_f32(xb, conv)
    assert yb.dtype == torch.bfloat16
    yb.backward(dy.to(torch.bfloat16))
    assert xb.grad.dtype == torch.bfloat16 and conv.weight.grad.dtype == torch.float32


def test_pools_f32_vs_torch():
    from dirhip.conv_f32 import global_avgpool_flat_f32, maxpool3x3s2_f32
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 24, 13, 10, generator=g)
    x[0, 0, 0, 0] = x[0, 0, 0, 1] = 5.0                       # a tie: the first maximum gets the gradient (torch semantics)
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = maxpool3x3s2_f32(xg)
    y.backward(dy.cuda())
    assert torch.equal(y.cpu(), ref.detach())
    assert_close(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-6, atol_scale=1e-7, msg="maxpool bwd")
    x7 = torch.randn(5, 40, 7, 7, generator=g)
    x7r = x7.clone().double().requires_grad_(True)
    r7 = F.avg_pool2d(x7r, 7).flatten(1)
    d7 = torch.randn(5, 40, generator=g)
    r7.backward(d7.double())
    x7g = x7.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y7 = global_avgpool_flat_f32(x7g)
    y7.backward(d7.cuda())
    assert_close(y7.detach().cpu().numpy(), r7.detach().numpy(), rtol=1e-6, atol_scale=1e-6, msg="avgpool")
    assert_close(x7g.grad.cpu().numpy(), x7r.grad.numpy(), rtol=1e-6, atol_scale=1e-7, msg="avgpool bwd")


def test_no_library_convolution_in_either_mode():
    """Neither the bf16 product path nor the float32 parity mode calls a library convolution / BatchNorm / pooling
    kernel: a profiler trace of one training step of each must not contain a MIOpen or ATen convolution kernel."""
    from torch.profiler import ProfilerActivity, profile
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    from dirhip.train_loop import resolve_loss, train_step
    torch.manual_seed(0)
    model = resnet50(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5,
                     sigma=2, momentum=0.9).cuda()
    x = torch.randn(4, 3, 224, 224, device="cuda")
    y = torch.tensor([[25.0], [31.0], [64.0], [25.0]], device="cuda")
    w = torch.ones(4, 1, device="cuda")
    for amp in (torch.bfloat16, None):
        eng = DataParallelEngine(model, amp_dtype=amp, channels_last=True)
        eng.train()
        opt = torch.optim.SGD(eng.parameters(), lr=1e-4)
        train_step(eng, opt, x, y, w, 0, resolve_loss("l1"))
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            train_step(eng, opt, x, y, w, 0, resolve_loss("l1"))
            torch.cuda.synchronize()
        names = {e.key for e in prof.key_averages()}
        bad = [n for n in names if any(t in n.lower() for t in ("miopen", "aten::convolution", "aten::cudnn", "aten::batch_norm",
                                                                "aten::native_batch_norm", "aten::max_pool2d", "aten::avg_pool2d",
                                                                "naive_conv", "igemm_", "implicitgemm"))
               and "conv_igemm_" not in n]                       # (our own kernels are called conv_igemm_[dma_]kernel)
        assert not bad, (amp, bad)
