"""-m gpu: fused BatchNorm(+residual)(+ReLU) HIP kernels vs a plain PyTorch fp32 reference of the same op
(F.batch_norm -> + residual -> relu, autograd backward), on ResNet-50's layer shapes, bf16 and f32, and the
whole dirhip ResNet-50 (fp32 mode) vs the reference's CPU forward golden."""
import numpy as np
import pytest
import torch

import variant_switches as VS  # tools/variant_switches.py: the product package has no setters (conftest puts tools/ on the path)
import torch.nn as nn
import torch.nn.functional as F

from conftest import assert_close
from dirhip import _lib as L

pytestmark = pytest.mark.gpu


def _ref(x, res, bn_w, bn_b, rm, rv, relu, training, momentum=0.1, eps=1e-5):
    y = F.batch_norm(x, rm, rv, bn_w, bn_b, training, momentum, eps)
    if res is not None:
        y = y + res
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,relu,has_res", [((8, 64, 56, 56), True, False), ((8, 256, 56, 56), True, True),
                                                ((4, 512, 28, 28), False, False), ((6, 1024, 14, 14), True, True),
                                                ((5, 2048, 7, 7), True, True), ((3, 64, 112, 112), True, False),
                                                ((2, 128, 9, 5), False, True), ((2, 8, 3, 2), True, False)])
def test_bn_act_train_fwd_bwd(dtype, shape, relu, has_res):
    from dirhip.bn import bn_act
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    n, c, h, w = shape
    x32 = (torch.randn(shape, device="cuda", generator=g) * 1.7 + 0.3).contiguous(memory_format=torch.channels_last)
    r32 = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last) if has_res else None
    dy32 = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    bn = nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, device="cuda", generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, device="cuda", generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(c, device="cuda", generator=g) * 0.1)
    # inputs rounded to the kernel dtype so both sides see identical values
    x = x32.to(dtype).requires_grad_(True)
    r = r32.to(dtype).requires_grad_(True) if has_res else None
    dy = dy32.to(dtype)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    y = bn_act(x, bn, relu=relu, residual=r)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    # fp32 reference of the same op
    xr = x.detach().float().requires_grad_(True)
    rr = r.detach().float().requires_grad_(True) if has_res else None
    wr, br = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    yr = _ref(xr, rr, wr, br, rm, rv, relu, True)
    yr.backward(dy.float())
    tol = dict(rtol=1e-5, atol_scale=2e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol_scale=4e-3)
    assert_close(y.detach().float().cpu().numpy(), yr.detach().cpu().numpy(), msg="y", **tol)
    assert_close(bn.running_mean.cpu().numpy(), rm.cpu().numpy(), rtol=1e-5, atol_scale=1e-6, msg="running_mean")
    assert_close(bn.running_var.cpu().numpy(), rv.cpu().numpy(), rtol=1e-5, atol_scale=1e-6, msg="running_var")
    assert int(bn.num_batches_tracked) == 1
    gt = dict(rtol=1e-4, atol_scale=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol_scale=1e-2)
    if dtype == torch.bfloat16 and relu:
        # the ReLU mask is taken from the bf16-rounded output: elements whose fp32 pre-activation is within bf16
        # round-off of 0 may flip; compare on the elements where both masks agree (all but a handful)
        agree = ((y.detach().float() > 0) == (yr.detach() > 0))
        assert agree.float().mean() > 0.999
    assert_close(x.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), msg="dx", **gt)
    if has_res:
        assert_close(r.grad.float().cpu().numpy(), rr.grad.cpu().numpy(), msg="dres", **gt)
    assert_close(bn.weight.grad.cpu().numpy(), wr.grad.cpu().numpy(), msg="dgamma", **gt)
    assert_close(bn.bias.grad.cpu().numpy(), br.grad.cpu().numpy(), msg="dbeta", **gt)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_act_eval(dtype):
    from dirhip.bn import bn_act
    g = torch.Generator(device="cuda").manual_seed(1)
    shape = (4, 256, 14, 14)
    x = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last).to(dtype)
    r = torch.randn(shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last).to(dtype)
    bn = nn.BatchNorm2d(256).cuda().eval()
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(256, device="cuda", generator=g) * 0.3)
        bn.running_var.copy_(torch.rand(256, device="cuda", generator=g) + 0.5)
        bn.weight.copy_(torch.rand(256, device="cuda", generator=g) + 0.5)
        y = bn_act(x, bn, relu=True, residual=r)
        yr = _ref(x.float(), r.float(), bn.weight, bn.bias, bn.running_mean, bn.running_var, True, False)
    tol = dict(rtol=1e-5, atol_scale=2e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol_scale=4e-3)
    assert_close(y.float().cpu().numpy(), yr.cpu().numpy(), **tol)
    assert int(bn.num_batches_tracked) == 0


def test_resnet50_forward_fp32_vs_reference_golden(golden):
    """dirhip ResNet-50 on the MI355X in float32 mode (hand-written exact-float32 MFMA convs + fused HIP BN nodes + HIP
    tail; no library kernel) vs the reference's CPU fp32 forward (golden): eval and train mode, same seeded init stream
    and input, at the north_star's 1e-5."""
    from dirhip.resnet import resnet50
    g = golden("resnet50_forward.npz")
    torch.manual_seed(1234)
    m = resnet50(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian",
                 ks=5, sigma=2, momentum=0.9).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(99)).cuda().contiguous(memory_format=torch.channels_last)
    m.eval()
    with torch.no_grad():
        p = m(x)
    assert_close(p.cpu().numpy(), g["ref_pred_eval"], rtol=1e-5, atol_scale=1e-5, msg="eval pred")
    m.train()
    with torch.no_grad():
        out = m(x, torch.tensor([[31.0], [64.0]], device="cuda"), 0)
    assert isinstance(out, tuple) and out[1].shape == (2, 2048)
    assert_close(out[0].cpu().numpy(), g["ref_pred_train"], rtol=1e-5, atol_scale=1e-5, msg="train pred")
    # B = 2 batch statistics amplify float32 rounding more than a real batch does: relative L2 1.1e-5, single elements
    # up to 6e-5 of the scale (the B = 64 comparison is tests/test_hip_step0_parity.py)
    enc, ref = out[1].double().cpu().numpy(), g["ref_enc_train"].astype(np.float64)
    assert np.linalg.norm(enc - ref) / np.linalg.norm(ref) <= 3e-5
    assert_close(enc, ref, rtol=1e-5, atol_scale=2e-4, msg="train encoding")


def test_maxpool3x3s2_matches_torch():
    from dirhip.pool import maxpool3x3s2
    g = torch.Generator(device="cuda").manual_seed(3)
    for shape in ((4, 64, 112, 112), (3, 64, 9, 7), (2, 8, 5, 6)):
        x = torch.relu(torch.randn(shape, device="cuda", generator=g)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        pool = nn.MaxPool2d(3, 2, 1)
        y = maxpool3x3s2(x, pool)
        dy = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y.backward(dy)
        xr = x.detach().float().requires_grad_(True)
        yr = pool(xr)
        yr.backward(dy.float())
        assert torch.equal(y.float(), yr)                                  # max of bf16 values is exact
        # ties (many exact zeros after ReLU) must route the gradient like torch: first maximum in scan order
        assert_close(x.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-2, atol_scale=1e-2, msg=str(shape))


def test_global_avgpool_matches_fp32_reference():
    from dirhip.pool import global_avgpool_flat
    g = torch.Generator(device="cuda").manual_seed(4)
    for shape in ((8, 2048, 7, 7), (3, 64, 7, 7)):
        x = torch.randn(shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        pool = nn.AvgPool2d(7, stride=1)
        y = global_avgpool_flat(x, pool)
        assert y.dtype == torch.float32 and y.shape == (shape[0], shape[1])
        dy = torch.randn(y.shape, device="cuda", generator=g)
        y.backward(dy)
        xr = x.detach().float().requires_grad_(True)
        yr = pool(xr).view(shape[0], -1)
        yr.backward(dy)
        assert_close(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-6, atol_scale=1e-6, msg="mean")
        assert x.grad.dtype == torch.bfloat16
        assert torch.equal(x.grad, xr.grad.to(torch.bfloat16))              # dy / 49 rounded once to bf16
    # a window that is not the whole map is refused loudly (no library fallback)
    from dirhip._lib import DirHipError
    x = torch.randn(2, 16, 9, 9, device="cuda").to(torch.bfloat16)
    with pytest.raises(DirHipError):
        global_avgpool_flat(x, nn.AvgPool2d(7, stride=1))


@pytest.mark.parametrize("deferred", [False, True])
def test_bn_join_matches_two_separate_batchnorms(deferred):
    """relu(bn(x) + bn_r(r)) in one apply pass vs BatchNorm(r) materialised first (fp32 torch reference for both)."""
    from dirhip.bn import bn_join
    g = torch.Generator(device="cuda").manual_seed(5)
    shape = (8, 256, 14, 14)
    x0 = torch.randn(shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r0 = (torch.randn(shape, device="cuda", generator=g) * 2 + 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bn, bn_r = nn.BatchNorm2d(256).cuda(), nn.BatchNorm2d(256).cuda()
    ref, ref_r = nn.BatchNorm2d(256).cuda(), nn.BatchNorm2d(256).cuda()
    with torch.no_grad():
        for m_, s_ in ((bn, 1), (bn_r, 2)):
            m_.weight.copy_(torch.rand(256, device="cuda", generator=g) + 0.5)
            m_.bias.copy_(torch.randn(256, device="cuda", generator=g) * 0.1)
        ref.load_state_dict(bn.state_dict()); ref_r.load_state_dict(bn_r.state_dict())
    x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
    y = bn_join(x, bn, None, r, bn_r, None, relu=True, defer_relu_grad=deferred)
    xf, rf = x0.float().requires_grad_(True), r0.float().requires_grad_(True)
    yf = torch.relu(ref(xf) + ref_r(rf))
    if deferred:                       # the consumer promises the masked gradient
        y._dir_relu_flag[0] = True
        y.backward(torch.where(y > 0, dy, torch.zeros_like(dy)))
    else:
        y.backward(dy)
    yf.backward(dy.float())
    assert_close(y.float().detach().cpu().numpy(), yf.detach().cpu().numpy(), rtol=1e-2, atol_scale=4e-3, msg="y")
    assert_close(x.grad.float().cpu().numpy(), xf.grad.cpu().numpy(), rtol=2e-2, atol_scale=8e-3, msg="dx")
    assert_close(r.grad.float().cpu().numpy(), rf.grad.cpu().numpy(), rtol=2e-2, atol_scale=8e-3, msg="dr")
    for a, b in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad), (bn_r.weight.grad, ref_r.weight.grad),
                 (bn_r.bias.grad, ref_r.bias.grad)):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-2, atol_scale=1e-2, msg="param grad")
    for a, b in ((bn.running_mean, ref.running_mean), (bn.running_var, ref.running_var), (bn_r.running_mean, ref_r.running_mean),
                 (bn_r.running_var, ref_r.running_var)):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol_scale=1e-4, msg="running stats")
    assert int(bn.num_batches_tracked) == 1 and int(bn_r.num_batches_tracked) == 1


@pytest.mark.parametrize("shape,dtype", [((8, 256, 56, 56), torch.bfloat16), ((6, 1024, 14, 14), torch.bfloat16), ((3, 2048, 7, 7), torch.bfloat16),
                                         ((2, 128, 9, 5), torch.float32)])
def test_bn_join_backward_one_pass_pair_is_bit_identical_to_two_backwards(shape, dtype):
    """dir_bn_bwd_join (one reduction + one apply pass for both BatchNorms of the join) vs two dir_bn_bwd(relu = 0) calls on the
    masked gradient: same row order, same accumulation order -> the same bits everywhere."""
    from dirhip import bn as B
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    c = shape[1]
    x0 = torch.randn(shape, device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    r0 = (torch.randn(shape, device="cuda", generator=g) * 2 + 0.5).to(dtype).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(shape, device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    got = {}
    for on in (True, False):
        prev = VS.set_join_bwd(on)
        try:
            bn, bn_r = nn.BatchNorm2d(c).cuda(), nn.BatchNorm2d(c).cuda()
            gg = torch.Generator(device="cuda").manual_seed(7)
            with torch.no_grad():
                for m_ in (bn, bn_r):
                    m_.weight.copy_(torch.rand(c, device="cuda", generator=gg) + 0.5)
                    m_.bias.copy_(torch.randn(c, device="cuda", generator=gg) * 0.1)
            x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            y = B.bn_join(x, bn, None, r, bn_r, None, relu=True, defer_relu_grad=True)
            y._dir_relu_flag[0] = True                                   # the consumer applied the ReLU backward
            y.backward(torch.where(y > 0, dy, torch.zeros_like(dy)))
            got[on] = [t.detach().clone() for t in (x.grad, r.grad, bn.weight.grad, bn.bias.grad, bn_r.weight.grad, bn_r.bias.grad)]
        finally:
            VS.set_join_bwd(prev)
    for a, b, name in zip(got[True], got[False], ("dx", "dr", "dgamma", "dbeta", "dgamma_r", "dbeta_r")):
        assert torch.equal(a, b), name


def test_stem_bn_relu_maxpool_matches_unfused():
    """Fused stem tail vs torch fp32 BatchNorm -> ReLU -> MaxPool2d(3, 2, 1)."""
    from dirhip.pool import bn_relu_maxpool
    g = torch.Generator(device="cuda").manual_seed(6)
    for shape in ((4, 64, 32, 32), (3, 64, 9, 7)):
        x0 = torch.randn(shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        bn, ref = nn.BatchNorm2d(64).cuda(), nn.BatchNorm2d(64).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(64, device="cuda", generator=g) + 0.5)
            bn.bias.copy_(torch.randn(64, device="cuda", generator=g) * 0.3)
            ref.load_state_dict(bn.state_dict())
        pool = nn.MaxPool2d(3, 2, 1)
        x = x0.clone().requires_grad_(True)
        y = bn_relu_maxpool(x, bn, pool)
        xr = x0.float().requires_grad_(True)
        yr = pool(torch.relu(ref(xr)))
        dy = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y.backward(dy)
        yr.backward(dy.float())
        assert_close(y.float().detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-2, atol_scale=4e-3, msg="y")

        def l2(a, b):
            return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
        # argmax ties / near-ties may route single gradient terms differently: compare in the L2 sense
        assert l2(x.grad.float(), xr.grad) < 3e-2, l2(x.grad.float(), xr.grad)
        assert l2(bn.weight.grad, ref.weight.grad) < 1e-2 and l2(bn.bias.grad, ref.bias.grad) < 1e-2
        assert_close(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), rtol=1e-4, atol_scale=1e-4, msg="running mean")
        assert_close(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), rtol=1e-4, atol_scale=1e-4, msg="running var")
        assert int(bn.num_batches_tracked) == 1


def test_stem_tail_backward_modes_bit_identical():
    """The stem tail's backward with the forward's xmax (streaming reduction + 2 x 2-block apply pass: dir_bn_relu_maxpool_*_xmax, what
    dirhip.pool runs) against the older pair without it (gather reduction + per-pixel apply pass: dir_bn_relu_maxpool_fwd / _bwd), even
    and odd map sizes: y, dx, dgamma, dbeta equal bit for bit. The form is chosen per call (xmax given or not), not by a switch."""
    from dirhip import pool as P
    g = torch.Generator(device="cuda").manual_seed(16)
    for shape in ((8, 64, 112, 112), (3, 64, 9, 7), (2, 64, 10, 15), (2, 32, 8, 8)):
        x0 = torch.randn(shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        c = shape[1]
        dy = None
        got = {}
        for use_xmax in (False, True):
            prev = VS.set_stem_tail_xmax(use_xmax)
            try:
                bn = nn.BatchNorm2d(c).cuda()
                with torch.no_grad():
                    bn.weight.copy_(torch.linspace(0.5, 1.5, c, device="cuda"))
                    bn.bias.copy_(torch.linspace(-0.4, 0.4, c, device="cuda"))
                x = x0.clone().requires_grad_(True)
                y = P.bn_relu_maxpool(x, bn, nn.MaxPool2d(3, 2, 1))
                if dy is None:
                    dy = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                y.backward(dy)
                got[use_xmax] = (y.detach().clone(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
            finally:
                VS.set_stem_tail_xmax(prev)
        for a, b, name in zip(got[True], got[False], ("y", "dx", "dgamma", "dbeta")):
            assert torch.equal(a, b), (shape, name)
