"""-m gpu: MFMA implicit-GEMM convolution (dir_conv_fwd) vs a plain PyTorch fp32 reference of the same op
(F.conv2d in float32 on the bf16-rounded inputs), including borders, strides, ragged M tails, and the fused
BatchNorm partial statistics."""
import numpy as np
import pytest
import torch

import variant_switches as VS  # tools/variant_switches.py: the product package has no setters (conftest puts tools/ on the path)
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu

CASES = [  # (N, Cin, H, W, Cout, k, stride, pad)
    (2, 64, 56, 56, 64, 1, 1, 0), (2, 64, 56, 56, 256, 1, 1, 0), (3, 256, 28, 28, 128, 1, 1, 0),
    (2, 256, 56, 56, 512, 1, 2, 0), (1, 1024, 14, 14, 2048, 1, 2, 0), (5, 2048, 7, 7, 512, 1, 1, 0),
    (2, 64, 56, 56, 64, 3, 1, 1), (2, 128, 56, 56, 128, 3, 2, 1), (3, 256, 14, 14, 256, 3, 1, 1),
    (2, 512, 7, 7, 512, 3, 1, 1), (1, 64, 9, 13, 192, 3, 1, 1), (7, 128, 5, 3, 64, 3, 2, 1), (1, 64, 1, 1, 64, 1, 1, 0),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_igemm_matches_fp32_reference(case):
    from dirhip.conv import conv2d_igemm
    n, cin, h, w, cout, k, stride, pad = case
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    # asymmetric weights (distinct per (co, ci, r, s)) so any operand transposition shows up
    wt = (torch.randn(cout, cin, k, k, device="cuda", generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16)
    wt = wt.contiguous(memory_format=torch.channels_last)
    y, stats = conv2d_igemm(x, wt, stride, pad, want_stats=True)
    ref = F.conv2d(x.float(), wt.float(), None, stride, pad)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    # bf16 output rounding: 2^-9 relative of the element + fp32 accumulation-order noise relative to the scale
    assert_close(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=4e-3, atol_scale=2e-3, msg=str(case))
    # fused statistics = sums over the rounded outputs
    yr = y.float().permute(0, 2, 3, 1).reshape(-1, cout).double()
    s = stats.double().sum(0).cpu().numpy()
    assert_close(s[0], yr.sum(0).cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sum")
    assert_close(s[1], (yr * yr).sum(0).cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sumsq")


def test_conv_igemm_identity_weights_asymmetric_input():
    """A = I check with an asymmetric input: the output must be the input, bit for bit (1x1, identity weights)."""
    from dirhip.conv import conv2d_igemm
    c = 128
    x = torch.arange(2 * c * 6 * 5, device="cuda").float().reshape(2, 6, 5, c).permute(0, 3, 1, 2) % 251
    x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.eye(c, device="cuda").reshape(c, c, 1, 1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = conv2d_igemm(x, w)
    assert torch.equal(y, x)


def test_conv_stats_feed_fused_batchnorm():
    """conv (MFMA) -> BatchNorm with the statistics taken from the conv epilogue == the same BatchNorm computing its
    own statistics from y, and == the fp32 reference chain; gradients flow through the custom conv node."""
    import torch.nn as nn
    from dirhip.bn import bn_act
    from dirhip.conv import conv_bn_input
    g = torch.Generator(device="cuda").manual_seed(5)
    conv = nn.Conv2d(128, 256, 3, 1, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    bn_a, bn_b = nn.BatchNorm2d(256).cuda(), nn.BatchNorm2d(256).cuda()
    x = torch.randn(6, 128, 14, 14, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    y, partial = conv_bn_input(x, conv, want_stats=True)
    assert partial is not None and partial.shape[1:] == (2, 256)
    out_a = bn_act(y, bn_a, relu=True, partial=partial)
    with torch.no_grad():
        out_b = bn_act(y.detach(), bn_b, relu=True)
    assert_close(out_a.float().detach().cpu().numpy(), out_b.float().cpu().numpy(), rtol=1e-2, atol_scale=4e-3)
    assert_close(bn_a.running_mean.cpu().numpy(), bn_b.running_mean.cpu().numpy(), rtol=1e-5, atol_scale=1e-6)
    assert_close(bn_a.running_var.cpu().numpy(), bn_b.running_var.cpu().numpy(), rtol=1e-5, atol_scale=1e-6)
    dy = torch.randn(out_a.shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out_a.backward(dy)
    # fp32 reference chain on the same bf16-rounded operands
    xr = x.detach().float().requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    outr = torch.relu(F.batch_norm(yr, None, None, torch.ones(256, device="cuda"), torch.zeros(256, device="cuda"), True, 0.1, 1e-5))
    outr.backward(dy.float())
    assert_close(out_a.float().detach().cpu().numpy(), outr.detach().cpu().numpy(), rtol=2e-2, atol_scale=1e-2, msg="out")
    # gradients pass through bf16 storage and a ReLU mask taken from bf16-rounded activations: a handful of
    # elements near the ReLU kink flip, so the check is norm-wise (relative L2 error) plus an outlier budget
    def l2(a, b):
        return float((a - b).norm() / b.norm())
    assert l2(x.grad.float(), xr.grad) < 2e-2, l2(x.grad.float(), xr.grad)
    assert l2(conv.weight.grad, wr.grad) < 2e-2, l2(conv.weight.grad, wr.grad)
    bad = ((x.grad.float() - xr.grad).abs() > 5e-2 * xr.grad.abs() + 2e-2 * xr.grad.abs().max()).float().mean().item()
    assert bad < 5e-3, bad
    assert conv.weight.grad.dtype == torch.float32


def test_weight_cache_follows_fused_optimizer_steps():
    """torch's fused Adam does not bump Tensor._version; the bf16 weight cache must still refresh after step()."""
    import torch.nn as nn
    from dirhip.conv import conv_bn_input
    conv = nn.Conv2d(64, 64, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    opt = torch.optim.Adam(conv.parameters(), lr=0.1, fused=True)
    x = torch.randn(2, 64, 8, 8, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y0, _ = conv_bn_input(x, conv, False)
    y0.float().sum().backward()
    opt.step()
    y1, _ = conv_bn_input(x, conv, False)
    ref = F.conv2d(x.float(), conv.weight.detach().to(torch.bfloat16).float())
    assert not torch.equal(y0, y1)
    assert_close(y1.float().detach().cpu().numpy(), ref.cpu().numpy(), rtol=4e-3, atol_scale=2e-3)


@pytest.mark.parametrize("case", CASES)
def test_conv_wgrad_matches_fp32_reference(case):
    from dirhip.conv import conv2d_wgrad
    n, cin, h, w, cout, k, stride, pad = case
    g = torch.Generator(device="cuda").manual_seed(sum(case) + 1)
    x = torch.randn(n, cin, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(n, cout, ho, wo, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = conv2d_wgrad(dy, x, k, stride, pad)
    wref = torch.zeros(cout, cin, k, k, device="cuda", requires_grad=True)
    F.conv2d(x.float(), wref, None, stride, pad).backward(dy.float())
    assert dw.shape == wref.grad.shape and dw.dtype == torch.float32
    assert_close(dw.cpu().numpy(), wref.grad.cpu().numpy(), rtol=2e-4, atol_scale=2e-4, msg=str(case))
    # deterministic split-K: bit-identical on a second run
    assert torch.equal(dw, conv2d_wgrad(dy, x, k, stride, pad))


@pytest.mark.parametrize("n,cin,cout,hw", [(7, 256, 1024, 14), (3, 1024, 256, 14), (5, 512, 128, 28), (33, 128, 128, 7), (1, 128, 256, 5), (64, 2048, 512, 7)])
def test_conv_wgrad_1x1_forms_agree(n, cin, cout, hw):
    """The three forms of the 1x1 / stride-1 weight gradient, chosen per launch (DIR_WGRAD_*): the transposing kernel, and the register-lean
    LDS-DMA + ds_read_b64_tr_b16 kernel with one / two LDS stages (the product's choice for 128-multiples of channels) — each against a
    float64 reference (ragged last K-step, one to hundreds of split-K ranges), each deterministic; the forced DMA forms refuse other geometries."""
    from dirhip import _lib as L
    from dirhip.conv import conv2d_wgrad
    g = torch.Generator(device="cuda").manual_seed(n + cin + cout + hw)
    x = torch.randn(n, cin, hw, hw, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cout, hw, hw, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref = torch.einsum("nohw,nihw->oi", dy.double(), x.double()).cpu().numpy()
    outs = {}
    for form in (L.WGRAD_AUTO, L.WGRAD_TRANSPOSE, L.WGRAD_DMA1, L.WGRAD_DMA2):
        dw = conv2d_wgrad(dy, x, 1, 1, 0, form=form)
        assert_close(dw.view(cout, cin).cpu().numpy(), ref, rtol=2e-5, atol_scale=2e-6, msg=f"form {form}")
        assert torch.equal(dw, conv2d_wgrad(dy, x, 1, 1, 0, form=form))
        outs[form] = dw
    assert torch.equal(outs[L.WGRAD_AUTO], outs[L.WGRAD_DMA2])                       # what the product runs for these shapes
    # not a 1x1 / stride-1 / 128-multiple geometry: the forced DMA forms are refused, AUTO takes the transposing kernel
    lib = L.lib()
    assert lib.dir_conv_wgrad_workspace(n, hw, hw, 64, cout, 1, 1, 1, 0, L.WGRAD_DMA2) == 0
    assert lib.dir_conv_wgrad_workspace(n, hw, hw, cin, cout, 3, 3, 1, 1, L.WGRAD_DMA1) == 0
    assert lib.dir_conv_wgrad_workspace(n, hw, hw, 64, cout, 1, 1, 1, 0, L.WGRAD_AUTO) > 0
    assert lib.dir_conv_wgrad_workspace(n, hw, hw, cin, cout, 1, 1, 1, 0, 7) == 0


def test_bottleneck_fused_shortcut_gradient_matches_unfused():
    """Identity-shortcut bottleneck: the gradient accumulation at the block input (dgrad(conv1) + d shortcut) fused into
    conv1's data-gradient kernel must give the same input/parameter gradients as autograd's eager add."""
    from dirhip import resnet as R
    torch.manual_seed(0)
    blk = R.Bottleneck(256, 64).cuda().to(memory_format=torch.channels_last)
    g = torch.Generator(device="cuda").manual_seed(11)
    x0 = torch.randn(8, 256, 14, 14, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(8, 256, 14, 14, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(fused):
        for p in blk.parameters():
            p.grad = None
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        x = x0.clone().requires_grad_(True)
        xin = x * 1.0                                        # non-leaf input like inside the network
        if fused:
            out = blk(xin)
        else:
            sc = xin
            y = R._conv_bn(xin, blk.conv1, blk.bn1, relu=True)
            y = R._conv_bn(y, blk.conv2, blk.bn2, relu=True)
            out = R._conv_bn(y, blk.conv3, blk.bn3, relu=True, residual=sc)
        out.backward(dy)
        return out.detach().float(), x.grad.float(), [p.grad.clone() for p in blk.parameters()]

    prev = VS.set_bn_bwd_fusion(False)      # (the BatchNorm sums stay in their own kernel on both sides: this test is about the edges)
    try:
        o1, gx1, gp1 = run(True)
        o2, gx2, gp2 = run(False)
    finally:
        VS.set_bn_bwd_fusion(prev)
    assert torch.equal(o1, o2)
    assert_close(gx1.cpu().numpy(), gx2.cpu().numpy(), rtol=1e-2, atol_scale=4e-3, msg="dx")
    for a, b in zip(gp1, gp2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("stride", [1, 2])
def test_projection_bottleneck_fused_shortcut_gradient_matches_unfused(stride):
    """Projection-shortcut bottleneck (first block of a stage): the block input feeds conv1 and the downsample conv; the
    sum of their data gradients, fused into conv1's data-gradient kernel, must match autograd's eager add."""
    import torch.nn as nn
    from dirhip import resnet as R
    torch.manual_seed(1)
    down = nn.Sequential(nn.Conv2d(128, 256, 1, stride=stride, bias=False), nn.BatchNorm2d(256))
    blk = R.Bottleneck(128, 64, stride=stride, downsample=down).cuda().to(memory_format=torch.channels_last)
    g = torch.Generator(device="cuda").manual_seed(12)
    ho = 16 // stride
    x0 = torch.randn(8, 128, 16, 16, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(8, 256, ho, ho, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(fused):
        for p in blk.parameters():
            p.grad = None
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        x = x0.clone().requires_grad_(True)
        xin = x * 1.0
        if fused:
            out = blk(xin)
        else:
            sc = R._conv_bn(xin, blk.downsample[0], blk.downsample[1], relu=False)
            y = R._conv_bn(xin, blk.conv1, blk.bn1, relu=True)
            y = R._conv_bn(y, blk.conv2, blk.bn2, relu=True)
            out = R._conv_bn(y, blk.conv3, blk.bn3, relu=True, residual=sc)
        out.backward(dy)
        return out.detach().float(), x.grad.float(), [p.grad.clone() for p in blk.parameters()]

    o1, gx1, gp1 = run(True)
    o2, gx2, gp2 = run(False)
    # the block's own forward also joins bn3 and the projection's BatchNorm in one pass (no bf16 rounding of the
    # normalised shortcut in between): equal up to that rounding
    # (outputs within a bf16 ulp of zero flip their ReLU mask, so gradients are compared in the L2 sense)
    assert_close(o1.cpu().numpy(), o2.cpu().numpy(), rtol=1e-2, atol_scale=4e-3, msg="out")

    def l2(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()

    # ~0.2 % of the outputs lie within the skipped rounding step of zero; each flip moves one gradient term by O(1):
    # expected relative L2 distance sqrt(0.002) ~ 4 %
    assert l2(gx1, gx2) < 8e-2, l2(gx1, gx2)
    for a, b in zip(gp1, gp2):
        assert l2(a, b) < 8e-2, l2(a, b)


def test_batched_weight_preparation_matches_single_layer_path():
    """One launch re-casts every registered layer after an optimizer step; operands must equal the per-layer kernel's."""
    import torch.nn as nn
    from dirhip import conv as C
    torch.manual_seed(2)
    convs = [nn.Conv2d(64, 128, 1, bias=False), nn.Conv2d(128, 128, 3, padding=1, bias=False),
             nn.Conv2d(128, 64, 3, stride=2, padding=1, bias=False)]
    convs = [c.cuda().to(memory_format=torch.channels_last) for c in convs]
    opt = torch.optim.SGD([p for c in convs for p in c.parameters()], lr=0.5)
    x = torch.randn(2, 64, 8, 8, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def fwd():
        y, _ = C.conv_bn_input(x, convs[0], False)
        y, _ = C.conv_bn_input(y, convs[1], False)
        y, _ = C.conv_bn_input(y, convs[2], False)
        return y

    fwd().float().square().mean().backward()
    before = [c._dir_w16.w16.clone() for c in convs]
    opt.step()                                                     # every layer stale at once -> batched refresh
    fwd()
    for c, old in zip(convs, before):
        st = c._dir_w16
        w = c.weight.detach()
        assert not torch.equal(st.w16, old)
        assert torch.equal(st.w16, w.to(torch.bfloat16))
        if st.rot_mode == 0:
            ref_rot = w.to(torch.bfloat16).flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
            assert torch.equal(st.w16_rot, ref_rot)
        else:
            # stride-2 3x3: four parity-class weights [Cin][taps][Cout], packed (0,0) (0,1) (1,0) (1,1); tap (dr, ds) of a
            # class is filter element (r, s) = (1 | 2,0)[a][dr], likewise s
            assert c.stride[0] == 2
            w16 = w.to(torch.bfloat16)
            pieces = []
            for a in (0, 1):
                for b in (0, 1):
                    rs = [1] if a == 0 else [2, 0]
                    ss = [1] if b == 0 else [2, 0]
                    sub = w16[:, :, rs][:, :, :, ss]                          # [Cout, Cin, R', S']
                    pieces.append(sub.permute(1, 2, 3, 0).contiguous().reshape(-1))   # [Cin][R'][S'][Cout]
            assert torch.equal(st.w16_rot, torch.cat(pieces))


def test_deferred_relu_backward_is_bit_identical():
    """relu(bn3 + shortcut) -> next block: the next conv1's data-gradient kernel applies that ReLU's backward on store
    (dir_conv_fwd_fused) and the BatchNorm backward skips its mask. Must be bit-identical to the unfused order."""
    from dirhip import resnet as R
    torch.manual_seed(3)
    blk1 = R.Bottleneck(256, 64).cuda().to(memory_format=torch.channels_last)
    blk2 = R.Bottleneck(256, 64).cuda().to(memory_format=torch.channels_last)
    g = torch.Generator(device="cuda").manual_seed(13)
    x0 = torch.randn(8, 256, 14, 14, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(8, 256, 14, 14, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    params = list(blk1.parameters()) + list(blk2.parameters())

    def run(defer):
        for p in params:
            p.grad = None
        for m in list(blk1.modules()) + list(blk2.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        x = x0.clone().requires_grad_(True)
        mid = blk1(x * 1.0)
        flag = mid._dir_relu_flag
        if not defer:
            del mid._dir_relu_flag
        out = blk2(mid)
        assert flag[0] == defer
        out.backward(dy)
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in params]

    prev = VS.set_bn_bwd_fusion(False)      # bn3's sums would otherwise move into conv1's epilogue only when the ReLU is deferred
    try:
        o1, gx1, gp1 = run(True)
        o2, gx2, gp2 = run(False)
    finally:
        VS.set_bn_bwd_fusion(prev)
    assert torch.equal(o1, o2)
    assert torch.equal(gx1, gx2)
    for a, b in zip(gp1, gp2):
        assert torch.equal(a, b)


def test_relu_bit_mask_is_bit_identical_to_the_tensor_mask():
    """The deferred ReLU backward reads its mask as one bit per element emitted by the forward (dir_bn_fwd_train_bits /
    dir_bn_apply_bits -> dir_conv_dgrad_ex) instead of the block output itself: same decisions, so identical gradients — over
    an identity block, a stride-1 projection pair and a stride-2 one (the join's bits), chained."""
    import torch.nn as nn
    from dirhip import bn as B
    from dirhip import resnet as R
    torch.manual_seed(4)
    down1 = nn.Sequential(nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256))
    down2 = nn.Sequential(nn.Conv2d(256, 512, 1, stride=2, bias=False), nn.BatchNorm2d(512))
    blocks = [R.Bottleneck(64, 64, downsample=down1), R.Bottleneck(256, 64), R.Bottleneck(256, 128, stride=2, downsample=down2),
              R.Bottleneck(512, 128)]
    blocks = [b.cuda().to(memory_format=torch.channels_last) for b in blocks]
    params = [p for b in blocks for p in b.parameters()]
    g = torch.Generator(device="cuda").manual_seed(14)
    x0 = torch.relu(torch.randn(6, 64, 28, 28, device="cuda", generator=g)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(6, 512, 14, 14, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(bits):
        prev = VS.set_relu_bits(bits)
        try:
            for p in params:
                p.grad = None
            for b in blocks:
                for m in b.modules():
                    if isinstance(m, nn.BatchNorm2d):
                        m.reset_running_stats()
            x = x0.clone().requires_grad_(True)
            y = x * 1.0
            seen = []
            for b in blocks:
                y = b(y)
                seen.append(hasattr(y, "_dir_relu_bits"))
            y.backward(dy)
        finally:
            VS.set_relu_bits(prev)
        return y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in params], seen

    y1, gx1, gp1, seen1 = run(True)
    y0, gx0, gp0, seen0 = run(False)
    assert all(seen1) and not any(seen0)
    assert torch.equal(y1, y0) and torch.equal(gx1, gx0)
    for a, b in zip(gp1, gp0):
        assert torch.equal(a, b)


def test_dgrad_join_adds_compact_stride2_gradient_at_even_pixels():
    """dir_conv_dgrad_join: conv + up2(compact) must equal the dense result plus the zero-upsampled compact tensor, bitwise."""
    from dirhip.conv import conv2d_igemm
    g = torch.Generator(device="cuda").manual_seed(31)
    n, cin, cout, h = 3, 64, 128, 12
    x = torch.randn(n, cin, h, h, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device="cuda", generator=g) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    compact = torch.randn(n, cout, h // 2, h // 2, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    mask = torch.randn(n, cout, h, h, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dense = conv2d_igemm(x, w, 1, 0)
    up = torch.zeros_like(dense)
    up[:, :, ::2, ::2] = compact
    ref = (dense.float() + up.float()).to(torch.bfloat16)
    got = conv2d_igemm(x, w, 1, 0, addend_s2=compact)
    assert torch.equal(got, ref)
    got_m = conv2d_igemm(x, w, 1, 0, addend_s2=compact, relu_mask=mask)
    assert torch.equal(got_m, torch.where(mask.float() > 0, ref, torch.zeros_like(ref)))
    # a dense AND a compact shortcut gradient at once (no ResNet-50 layer does; the store loop's general form): dense first, each
    # addition rounded to bf16 like an eager add kernel
    dense_add = torch.randn(n, cout, h, h, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    step1 = (dense.float() + dense_add.float()).to(torch.bfloat16)
    ref2 = step1.clone()
    ref2[:, :, ::2, ::2] = (step1[:, :, ::2, ::2].float() + compact.float()).to(torch.bfloat16)
    got2 = conv2d_igemm(x, w, 1, 0, addend=dense_add, addend_s2=compact)
    assert torch.equal(got2, ref2)


@pytest.mark.parametrize("shape", [(4, 3, 224, 224), (2, 3, 64, 40), (3, 3, 18, 8)])
def test_stem_conv_matches_fp32_reference(shape):
    """dir_stem_conv_fwd (7x7/2, 3 -> 64) vs F.conv2d on the bf16-rounded operands in fp32; statistics vs the stored output."""
    import torch.nn as nn
    from dirhip.conv import stem_conv, stem_conv_ok
    torch.manual_seed(7)
    conv = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).cuda().to(memory_format=torch.channels_last)
    g = torch.Generator(device="cuda").manual_seed(41)
    x = torch.randn(shape, device="cuda", generator=g)
    assert stem_conv_ok(x, conv)
    y, stats = stem_conv(x, conv, want_stats=True)
    ref = F.conv2d(x.to(torch.bfloat16).float(), conv.weight.detach().to(torch.bfloat16).float(), None, 2, 3)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    assert_close(y.float().detach().cpu().numpy(), ref.cpu().numpy(), rtol=4e-3, atol_scale=2e-3, msg="y")
    yf = y.detach().double()
    tot = stats.double().sum(0)
    assert_close(tot[0].cpu().numpy(), yf.sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sum")
    assert_close(tot[1].cpu().numpy(), (yf * yf).sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sumsq")
    # weight gradient flows (library kernel), and the packed weights follow an optimizer step
    y.float().square().mean().backward()
    wr = conv.weight.detach().clone().requires_grad_(True)
    F.conv2d(x.to(torch.bfloat16).float(), wr, None, 2, 3).square().mean().backward()
    rel = ((conv.weight.grad - wr.grad).norm() / wr.grad.norm()).item()
    assert rel < 3e-2, rel
    opt = torch.optim.SGD(conv.parameters(), lr=1.0)
    opt.step()
    y2, _ = stem_conv(x, conv, want_stats=False)
    ref2 = F.conv2d(x.to(torch.bfloat16).float(), conv.weight.detach().to(torch.bfloat16).float(), None, 2, 3)
    assert_close(y2.float().detach().cpu().numpy(), ref2.cpu().numpy(), rtol=4e-3, atol_scale=2e-3, msg="y after step")


@pytest.mark.parametrize("shape", [(2, 128, 56, 56, 128), (3, 64, 8, 12, 192), (1, 256, 2, 2, 64)])
def test_stride2_3x3_data_gradient_by_parity_classes(shape):
    """dir_conv_dgrad_s2 (four stride-1 launches, one per output-pixel parity class) vs autograd of F.conv2d in fp32."""
    import torch.nn as nn
    from dirhip.conv import conv_bn_input
    n, cin, h, w, cout = shape
    torch.manual_seed(9)
    conv = nn.Conv2d(cin, cout, 3, stride=2, padding=1, bias=False).cuda().to(memory_format=torch.channels_last)
    g = torch.Generator(device="cuda").manual_seed(51)
    x0 = torch.randn(n, cin, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = x0.clone().requires_grad_(True)
    y, _ = conv_bn_input(x, conv, want_stats=False)
    assert conv._dir_w16.rot_mode == 1
    dy = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    xr = x0.float().requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    F.conv2d(xr, wr, None, 2, 1).backward(dy.float())
    assert x.grad.shape == xr.grad.shape and x.grad.dtype == torch.bfloat16
    assert_close(x.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-2, atol_scale=4e-3, msg="dx")
    rel = ((conv.weight.grad - wr.grad).norm() / wr.grad.norm()).item()
    assert rel < 1e-2, rel


def test_batched_weight_gradient_reduction_is_bit_identical_and_one_launch():
    """Round 5: inside a backward pass the split-K partials of every convolution layer are reduced by ONE launch in an autograd end-of-pass
    callback (dir_conv_wgrad_reduce_batched) instead of one launch per layer. Same gradients bit for bit as the per-layer form, `.grad`
    complete when backward() returns, and the profiler sees one reduction kernel per step instead of 52."""
    from torch.profiler import ProfilerActivity, profile
    from dirhip import conv as C
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    from dirhip.train_loop import resolve_loss

    def grads(batched):
        prev = VS.set_wgrad_batched_reduce(batched)
        try:
            torch.manual_seed(0)
            model = resnet50(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9).cuda()
            eng = DataParallelEngine(model, amp_dtype=torch.bfloat16, channels_last=True)
            eng.train()
            g = torch.Generator(device="cuda").manual_seed(1)
            x = torch.randn(8, 3, 224, 224, device="cuda", generator=g)
            y = torch.tensor([[25.0], [31.0], [64.0], [25.0]] * 2, device="cuda")
            w = torch.ones(8, 1, device="cuda")
            loss_fn = resolve_loss("l1")
            out = []
            names = []
            for it in range(2):                               # second pass: cached table, persistent workspaces
                eng.zero_grad(set_to_none=True)
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    loss = loss_fn(eng(x, y, 2)[0], y, w)
                    loss.backward()
                    got = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}      # read right after backward()
                    torch.cuda.synchronize()
                out.append(got)
                names = [(e.key, e.count) for e in prof.key_averages()]
            return out, names
        finally:
            VS.set_wgrad_batched_reduce(prev)
    a, names_a = grads(False)
    b, names_b = grads(True)
    for ga, gb in zip(a, b):
        assert ga.keys() == gb.keys() and len(ga) == 161
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k
    per_layer = sum(c for k, c in names_a if "conv_wgrad_reduce_kernel" in k)
    batched = sum(c for k, c in names_b if "conv_wgrad_reduce_batched_kernel" in k)
    assert per_layer == 52 and batched == 1 and not any("conv_wgrad_reduce_kernel" in k for k, _ in names_b), (names_a, names_b)
