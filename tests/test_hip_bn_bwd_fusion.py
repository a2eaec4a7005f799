"""-m gpu: the BatchNorm-backward reduction fused into the data-gradient kernel that produces the BatchNorm's ``dout``
(``dir_conv_dgrad_bnstats`` / ``dir_conv_dgrad_s2_bnstats`` + ``dir_bn_bwd_partials``; host side ``bn.BwdLink``) against the
plain three-pass ``dir_bn_bwd`` it replaces, and against float64 sums of the stored gradient.

Replaces the BatchNorm2d backward of imdb-wiki-dir/resnet.py:45-50 (bn1 / bn2 / bn3 of a Bottleneck).
"""
import numpy as np
import pytest
import torch

import variant_switches as VS  # tools/variant_switches.py: the product package has no setters (conftest puts tools/ on the path)

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _link(bnx, c, g, recompute):
    from dirhip.bn import BwdLink
    link = BwdLink()
    link.x = bnx
    link.gamma = torch.rand(c, device="cuda", generator=g) + 0.5
    link.beta = torch.randn(c, device="cuda", generator=g) * 0.3
    link.mean = torch.randn(c, device="cuda", generator=g) * 0.2
    link.rstd = torch.rand(c, device="cuda", generator=g) + 0.6
    link.recompute_mask = recompute
    return link


def _expected_sums(y, link):
    """float64 per-channel sums of g and g * x from the STORED bf16 gradient, mask as the forward decided it (float32 x*a+b > 0
    with a, b rounded from float64 — dir_bn.hip bn_mask_coef)."""
    gf = y.float()
    xf = link.x.float()
    if link.recompute_mask:
        a = (link.gamma.double() * link.rstd.double()).float().view(1, -1, 1, 1)
        b = (link.beta.double() - link.mean.double() * link.gamma.double() * link.rstd.double()).float().view(1, -1, 1, 1)
        gf = torch.where(xf * a + b > 0, gf, torch.zeros_like(gf))
    return gf.double().sum((0, 2, 3)), (gf.double() * xf.double()).sum((0, 2, 3))


@pytest.mark.parametrize("recompute", [True, False], ids=["relu_mask_recomputed", "no_mask"])
@pytest.mark.parametrize("n,cy,cx,hw,k,extras", [
    (8, 256, 64, 56, 1, ""),             # conv3 data gradient of layer1 -> bn2 sums (64-wide tile, K loop of 4)
    (8, 64, 64, 56, 3, ""),              # conv2 data gradient -> bn1 sums (3x3, register-staged loop)
    (4, 512, 512, 7, 3, ""),             # 3x3 with 72 K-steps: the LDS-DMA loop shares the epilogue
    (8, 64, 256, 56, 1, "addend+mask"),  # conv1 data gradient + shortcut gradient + deferred ReLU -> previous bn3 sums
    (3, 128, 512, 28, 1, "addend"),
    (5, 1024, 256, 14, 1, ""),           # M = 980: last 128-row tile is partial
    (5, 64, 256, 14, 1, "addend+mask"),  # ... with every fused operand: the rows beyond M fall outside the buffer descriptors' range
    (3, 128, 128, 14, 3, "addend"),      # patch-staged 3x3: every tile is 98 of 128 rows
])
def test_dgrad_epilogue_sums_vs_float64(n, cy, cx, hw, k, extras, recompute):
    from dirhip.conv import conv2d_igemm
    if "mask" in extras and recompute:
        pytest.skip("a BatchNorm behind relu(. + residual) never recomputes its mask")
    g = torch.Generator(device="cuda").manual_seed(n * 1000 + cy + cx + hw + k)
    dy = _cl((torch.randn(n, cy, hw, hw, device="cuda", generator=g) * 0.5).to(torch.bfloat16))
    w = _cl((torch.randn(cx, cy, k, k, device="cuda", generator=g) / np.sqrt(cy * k * k)).to(torch.bfloat16))
    bnx = _cl((torch.randn(n, cx, hw, hw, device="cuda", generator=g) * 1.3 + 0.2).to(torch.bfloat16))
    addend = _cl(torch.randn(n, cx, hw, hw, device="cuda", generator=g).to(torch.bfloat16)) if "addend" in extras else None
    mask = _cl(torch.randn(n, cx, hw, hw, device="cuda", generator=g).to(torch.bfloat16)) if "mask" in extras else None
    link = _link(bnx, cx, g, recompute)
    y = conv2d_igemm(dy, w, 1, k // 2, addend=addend, relu_mask=mask, bn_link=link)
    y_plain = conv2d_igemm(dy, w, 1, k // 2, addend=addend, relu_mask=mask)
    assert torch.equal(y, y_plain)                                   # the stored gradient is untouched by the fusion
    # the narrower entry point (no bit mask argument) is the same launch
    from dirhip.conv import _bn_link_args
    y2 = torch.empty_like(y)
    part2 = torch.empty_like(link.partial)
    L0 = __import__("dirhip._lib", fromlist=["lib"])
    L0.check(L0.lib().dir_conv_dgrad_bnstats(L0.ptr(dy), L0.ptr(w), L0.ptr(addend), None, L0.ptr(mask), L0.ptr(y2), n, hw, hw, cy, cx, k, k, k // 2,
                                             *_bn_link_args(link), L0.ptr(part2), part2.shape[0], L0.stream_ptr(dy.device)), "dir_conv_dgrad_bnstats")
    assert torch.equal(y2, y) and torch.equal(part2, link.partial)
    part = link.partial
    from dirhip import _lib as L
    assert part is not None and part.shape == (L.lib().dir_conv_plan_rows(n, hw, hw, cy, cx, k, k, 1, k // 2, 0, 0), 2, cx)
    s0, s1 = _expected_sums(y, link)
    got = part.double().sum(0)
    # per-tile float32 accumulation of <= 128 terms: 1e-5 of the scale of the column sums
    assert_close(got[0].cpu().numpy(), s0.cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sum g")
    assert_close(got[1].cpu().numpy(), s1.cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sum g*x")


@pytest.mark.parametrize("recompute", [True, False], ids=["relu_mask_recomputed", "no_mask"])
def test_stride2_dgrad_epilogue_sums_vs_float64(recompute):
    """The 3x3 / stride-2 data gradient runs as four parity-class launches with scattered rows: one block of partial rows per class."""
    from dirhip import _lib as L
    from dirhip.conv import _bn_link_args
    n, cy, cx, ho = 6, 128, 128, 14
    g = torch.Generator(device="cuda").manual_seed(77)
    conv = torch.nn.Conv2d(cx, cy, 3, stride=2, padding=1, bias=False).cuda()
    from dirhip.conv import _prepared
    w16, wcls = _prepared(conv)
    dy = _cl((torch.randn(n, cy, ho, ho, device="cuda", generator=g) * 0.5).to(torch.bfloat16))
    bnx = _cl((torch.randn(n, cx, 2 * ho, 2 * ho, device="cuda", generator=g) * 1.3 + 0.2).to(torch.bfloat16))
    link = _link(bnx, cx, g, recompute)
    dx = torch.empty_like(bnx)
    dx_plain = torch.empty_like(bnx)
    rows = L.lib().dir_conv_stats_rows(n, ho, ho)
    part = torch.empty(4 * rows, 2, cx, dtype=torch.float32, device="cuda")
    st = L.stream_ptr(dy.device)
    L.check(L.lib().dir_conv_dgrad_s2_bnstats(L.ptr(dy), L.ptr(wcls), L.ptr(dx), n, ho, ho, cy, cx, *_bn_link_args(link), L.ptr(part), part.shape[0], st), "s2 bnstats")
    # a list sized for another tiling is refused, not overrun
    assert L.lib().dir_conv_dgrad_s2_bnstats(L.ptr(dy), L.ptr(wcls), L.ptr(dx), n, ho, ho, cy, cx, *_bn_link_args(link), L.ptr(part), rows, st) == -1
    L.check(L.lib().dir_conv_dgrad_s2(L.ptr(dy), L.ptr(wcls), L.ptr(dx_plain), n, ho, ho, cy, cx, st), "s2")
    assert torch.equal(dx, dx_plain)
    # ... and it is the data gradient of the stride-2 convolution (float32 reference on the bf16 operands)
    ref = torch.nn.grad.conv2d_input(bnx.shape, w16.float(), dy.float(), stride=2, padding=1)
    assert_close(dx.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol_scale=4e-3, msg="dx")
    s0, s1 = _expected_sums(dx, link)
    got = part.double().sum(0)
    assert_close(got[0].cpu().numpy(), s0.cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sum g")
    assert_close(got[1].cpu().numpy(), s1.cpu().numpy(), rtol=1e-5, atol_scale=1e-5, msg="sum g*x")


@pytest.mark.parametrize("shape,relu", [((8, 64, 56, 56), True), ((8, 256, 56, 56), False), ((6, 1024, 14, 14), False),
                                         ((5, 512, 7, 7), True), ((32, 64, 56, 56), True)])
def test_bn_bwd_from_partials_equals_three_pass_backward(shape, relu):
    """dir_bn_bwd_partials fed per-128-row float32 sums (what the conv epilogue emits) vs dir_bn_bwd on the same tensors."""
    from dirhip import _lib as L
    from dirhip import bn as B
    n, c, h, w = shape
    m = n * h * w
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = _cl((torch.randn(shape, device="cuda", generator=g) * 1.7 + 0.3).to(torch.bfloat16))
    dout = _cl(torch.randn(shape, device="cuda", generator=g).to(torch.bfloat16))
    gamma = torch.rand(c, device="cuda", generator=g) + 0.5
    beta = torch.randn(c, device="cuda", generator=g) * 0.2
    xf = x.float().permute(0, 2, 3, 1).reshape(m, c)
    mean = xf.mean(0)
    rstd = 1.0 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5)
    gf = dout.float().permute(0, 2, 3, 1).reshape(m, c)
    if relu:
        a = (gamma.double() * rstd.double()).float()
        b = (beta.double() - mean.double() * gamma.double() * rstd.double()).float()
        gf = torch.where(xf * a + b > 0, gf, torch.zeros_like(gf))
    rows = (m + 127) // 128
    pad = rows * 128 - m
    gp = torch.cat([gf, gf.new_zeros(pad, c)]).view(rows, 128, c)
    xp = torch.cat([xf, xf.new_zeros(pad, c)]).view(rows, 128, c)
    part = torch.stack([gp.sum(1), (gp * xp).sum(1)], 1).contiguous()                    # [rows][2][C] float32
    code = L.DIR_BF16
    ws = torch.empty(L.lib().dir_bn_workspace(code, m, c), dtype=torch.uint8, device="cuda")
    st = L.stream_ptr(x.device)
    out = {}
    for which in ("plain", "partials"):
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        if which == "plain":
            L.check(L.lib().dir_bn_bwd(L.ptr(dout), L.ptr(x), None, L.ptr(dx), None, code, m, c, L.ptr(gamma), L.ptr(beta), L.ptr(mean),
                                       L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), int(relu), L.ptr(ws), ws.numel(), st), "dir_bn_bwd")
        else:
            L.check(L.lib().dir_bn_bwd_partials(L.ptr(dout), L.ptr(x), L.ptr(dx), code, m, c, L.ptr(gamma), L.ptr(beta), L.ptr(mean),
                                                L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta), int(relu), L.ptr(part), rows, L.ptr(ws),
                                                ws.numel(), st), "dir_bn_bwd_partials")
        out[which] = (dx.float().cpu().numpy(), dgamma.cpu().numpy(), dbeta.cpu().numpy())
    assert_close(out["partials"][1], out["plain"][1], rtol=1e-5, atol_scale=1e-5, msg="dgamma")
    assert_close(out["partials"][2], out["plain"][2], rtol=1e-5, atol_scale=1e-5, msg="dbeta")
    # dx = a*g + p*x + q in float32 from coefficients that agree to ~1e-6, rounded to bf16: at most one bf16 ulp apart
    assert_close(out["partials"][0], out["plain"][0], rtol=8e-3, atol_scale=1e-4, msg="dx")
    assert np.mean(out["partials"][0] != out["plain"][0]) < 2e-2


def _kernel_counts(fn):
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    counts = {}
    for e in prof.events():
        counts[e.name] = counts.get(e.name, 0) + 1
    return counts


def test_in_situ_sums_vs_float64_and_vs_three_pass(monkeypatch):
    """Wiring check inside a real training step: at every BatchNorm backward that received its sums from a conv epilogue, the
    sums are recomputed in float64 from the very tensors the node holds (dout as delivered, saved x, mask rule), and dgamma /
    dbeta of BOTH forms (epilogue sums, three-pass kernel) are compared with that truth. The two forms differ from each other
    by no more than each differs from float64: summation order, nothing else."""
    from dirhip import _lib as L
    from dirhip import bn as B
    from dirhip import resnet as R
    records = []
    orig = B._BNActFn.backward

    def checked(ctx, dout):
        link = getattr(ctx, "link", None)
        part = None if link is None else link.partial
        if part is not None:
            x, gamma, beta, y, mean, rstd = ctx.saved_tensors
            deferred = ctx.deferred is not None and ctx.deferred[0]
            assert deferred or not ctx.has_res                      # (behind relu(. + residual) only with the ReLU backward claimed)
            d = B._nhwc(dout)
            n, c, h, w = x.shape
            m = n * h * w
            gf = d.float()
            xf = x.float()
            relu = ctx.relu and not deferred
            if relu:
                a = (gamma.double() * rstd.double()).float().view(1, -1, 1, 1)
                b = (beta.double() - mean.double() * gamma.double() * rstd.double()).float().view(1, -1, 1, 1)
                gf = torch.where(xf * a + b > 0, gf, torch.zeros_like(gf))
            s0, s1 = gf.double().sum((0, 2, 3)), (gf.double() * xf.double()).sum((0, 2, 3))
            a0, a1 = gf.double().abs().sum((0, 2, 3)), (gf.double() * xf.double()).abs().sum((0, 2, 3))
            got = part.double().sum(0)
            e_sum = max(float(((got[0] - s0).abs() / a0.clamp_min(1e-30)).max()), float(((got[1] - s1).abs() / a1.clamp_min(1e-30)).max()))
            dg_true = rstd.double() * (s1 - mean.double() * s0)
            # three-pass form on the same tensors
            code = L.DIR_BF16
            ws = torch.empty(L.lib().dir_bn_workspace(code, m, c), dtype=torch.uint8, device=x.device)
            dx3 = torch.empty_like(x)
            dg3, db3 = torch.empty(c, device=x.device), torch.empty(c, device=x.device)
            L.check(L.lib().dir_bn_bwd(L.ptr(d), L.ptr(x), None, L.ptr(dx3), None, code, m, c, L.ptr(gamma), L.ptr(beta), L.ptr(mean),
                                       L.ptr(rstd), L.ptr(dg3), L.ptr(db3), int(relu), L.ptr(ws), ws.numel(), L.stream_ptr(x.device)), "dir_bn_bwd")
        out = orig(ctx, dout)
        if part is not None:
            dgf, dbf = out[1], out[2]
            scale_g = (rstd.double() * (a1 + mean.double().abs() * a0)).clamp_min(1e-30)      # what the cancellation is measured against
            rec = {"shape": tuple(x.shape), "relu": bool(relu), "e_sum": e_sum,
                   "dgamma_fused": float(((dgf.double() - dg_true).abs() / scale_g).max()),
                   "dgamma_3pass": float(((dg3.double() - dg_true).abs() / scale_g).max()),
                   "dbeta_fused": float(((dbf.double() - s0).abs() / a0.clamp_min(1e-30)).max()),
                   "dbeta_3pass": float(((db3.double() - s0).abs() / a0.clamp_min(1e-30)).max()),
                   "dx_diff_frac": float((out[0] != dx3).float().mean()),
                   "dx_rel_l2": float((out[0].double() - dx3.double()).norm() / dx3.double().norm().clamp_min(1e-300))}
            records.append(rec)
        return out

    monkeypatch.setattr(B._BNActFn, "backward", staticmethod(checked))
    torch.manual_seed(3)
    model = R.resnet50(fds=False, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5,
                       sigma=2, momentum=0.9).cuda().to(memory_format=torch.channels_last)
    model.train()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = _cl(torch.randn(16, 3, 224, 224, device="cuda", generator=g))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
    out.float().square().mean().backward()
    assert len(records) == 43, len(records)
    import json
    import os
    from conftest import records_dir
    json.dump(records, open(os.path.join(records_dir(), "bn_bwd_fusion_in_situ.json"), "w"), indent=1)
    worst = {k: max(r[k] for r in records) for k in ("e_sum", "dgamma_fused", "dgamma_3pass", "dbeta_fused", "dbeta_3pass", "dx_rel_l2", "dx_diff_frac")}
    # float32 accumulation of <= 128 terms per tile, float64 across tiles: 2e-6 of the sum of magnitudes
    assert worst["e_sum"] <= 2e-6, worst
    assert worst["dgamma_fused"] <= 2e-6 and worst["dbeta_fused"] <= 2e-6, worst
    # ... and the epilogue form is at least as accurate as the three-pass kernel (which accumulates long float32 chains per thread)
    assert worst["dgamma_fused"] <= max(2e-6, 2 * worst["dgamma_3pass"]), worst


def test_whole_model_fused_vs_three_pass_bn_backward():
    """ResNet-50 training step (bf16 product path) with the fusion on and off: same forward bit for bit, gradients equal up to the
    summation order of the BatchNorm reductions; and the reduction kernel really is gone from 43 of the 52 BatchNorm backwards
    (bn1 / bn2 of every block, bn3 of the 11 identity blocks that feed another block; the 4 two-BatchNorm joins and the last block
    keep the three-pass form)."""
    from dirhip import resnet as R
    torch.manual_seed(3)
    model = R.resnet50(fds=False, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5,
                       sigma=2, momentum=0.9).cuda().to(memory_format=torch.channels_last)
    model.train()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = _cl(torch.randn(16, 3, 224, 224, device="cuda", generator=g))

    def step():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        model.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
        out.float().square().mean().backward()
        return out.detach().float().clone()

    res = {}
    counts = {}
    for fused in (True, False):
        prev = VS.set_bn_bwd_fusion(fused)
        try:
            counts[fused] = _kernel_counts(step)
            out = step()
        finally:
            VS.set_bn_bwd_fusion(prev)
        res[fused] = (out, {n: p.grad.detach().double().clone() for n, p in model.named_parameters()})

    def n_partial(c):                               # (reduction passes of single BatchNorms, of the four joins' pairs)
        return (sum(v for k, v in c.items() if "bn_bwd_partial_kernel" in k), sum(v for k, v in c.items() if "bn_bwd_join_partial_kernel" in k))
    assert n_partial(counts[False]) == (44, 4) and n_partial(counts[True]) == (1, 4), (n_partial(counts[False]), n_partial(counts[True]))
    assert torch.equal(res[True][0], res[False][0])
    rels = {n: float((res[True][1][n] - res[False][1][n]).norm() / res[False][1][n].norm().clamp_min(1e-300)) for n in res[True][1]}
    worst = sorted(rels.items(), key=lambda kv: -kv[1])[:5]
    by_stage = {s: max(v for k, v in rels.items() if k.startswith(s)) for s in ("layer4", "layer3", "layer2", "layer1")}
    # Per BatchNorm the two forms agree to 1e-6 (test_in_situ_sums_vs_float64_and_vs_three_pass: both within 1e-8 of the float64
    # sums, dx within 2e-6 on identical inputs). Downstream, every bf16 rounding stage turns a relative perturbation d << 2^-8
    # into sqrt(d * 2^-8) (a fraction d / 2^-8 of the elements round the other way, each by one ulp = 2^-8): 1e-6 -> 6e-5 ->
    # 5e-4 -> 1.4e-3 -> ... with the bf16 rounding floor 2^-8 = 0.4 % as its fixed point, reached within one Bottleneck; from
    # there the backward of the random-init network amplifies towards the stem as it does for any bf16 rounding difference.
    # Measured: layer4 5e-3 (= the floor), layer1 2e-2, stem 7e-2.
    assert by_stage["layer4"] <= 2e-2 and by_stage["layer3"] <= 5e-2, (by_stage, worst)
    assert max(rels.values()) <= 0.3 and float(np.median(list(rels.values()))) <= 3e-2, (by_stage, worst)
    # last BatchNorm on the backward path whose sums come from a conv epilogue: reduction-order agreement
    assert rels["layer4.2.bn2.weight"] <= 1e-4 and rels["layer4.2.bn2.bias"] <= 1e-4, (rels["layer4.2.bn2.weight"], rels["layer4.2.bn2.bias"])
    import json
    import os
    from conftest import records_dir
    json.dump({"by_stage_max_rel_l2": by_stage, "median": float(np.median(list(rels.values()))), "worst": worst},
              open(os.path.join(records_dir(), "bn_bwd_fusion_whole_model.json"), "w"), indent=1)
