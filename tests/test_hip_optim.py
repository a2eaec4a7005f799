"""-m gpu: dirhip.optim.Adam (ONE launch for all parameters: dir_adam_step, fused with the bf16 weight-operand preparation) against
torch.optim.Adam's single-tensor implementation on the same gradients, for the whole ResNet-50 parameter set: parameters and
state after three steps, the state_dict layout, and the convolution operands the kernel rewrote (must equal
dir_conv_prep_weights of the updated master weights bit for bit, and be what the next forward actually uses)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model_with_grads():
    from dirhip.parallel import DataParallelEngine
    from dirhip.resnet import resnet50
    torch.manual_seed(3)
    model = resnet50(fds=False, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2,
                     momentum=0.9).cuda()
    eng = DataParallelEngine(model, amp_dtype=torch.bfloat16, channels_last=True)
    eng.train()
    x = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(4)).cuda()
    eng(x).sum().backward()
    return model, eng, x


@pytest.mark.parametrize("weight_decay", [0.0, 1e-4])
def test_adam_matches_torch_single_tensor_adam_and_rewrites_conv_operands(weight_decay):
    from dirhip import _lib as L
    from dirhip import conv as C
    from dirhip.optim import Adam
    model, eng, x = _model_with_grads()
    ref_params = [p.detach().clone().requires_grad_(True) for p in model.parameters()]
    for rp, p in zip(ref_params, model.parameters()):
        rp.grad = p.grad.detach().clone()
    ours = Adam(model.parameters(), lr=1e-3, weight_decay=weight_decay)
    theirs = torch.optim.Adam(ref_params, lr=1e-3, weight_decay=weight_decay, foreach=False, fused=False)
    for _ in range(3):
        theirs.step()                                             # (every optimizer.step() opens a new weight-cache generation: ours last)
        ours.step()
    torch.cuda.synchronize()
    worst = 0.0
    for (n, p), rp in zip(model.named_parameters(), ref_params):
        d = (p.detach().double() - rp.detach().double()).abs().max().item()
        worst = max(worst, d / max(rp.detach().abs().max().item(), 1e-12))
        so, st = ours.state[p], theirs.state[rp]
        assert float(so["step"]) == float(st["step"]) == 3.0
        assert torch.allclose(so["exp_avg"], st["exp_avg"], rtol=1e-6, atol=1e-12), n
        assert torch.allclose(so["exp_avg_sq"], st["exp_avg_sq"], rtol=1e-6, atol=1e-20), n
    assert worst <= 2e-6, worst                                   # float32 round-off of one or two operations (torch fuses nothing here)
    sd_o, sd_t = ours.state_dict(), theirs.state_dict()
    assert sd_o["param_groups"][0].keys() == sd_t["param_groups"][0].keys()
    assert all(set(v.keys()) == {"step", "exp_avg", "exp_avg_sq"} for v in sd_o["state"].values())
    # the bf16 operands of every conv layer = dir_conv_prep_weights of the UPDATED master weight, and valid for the next forward
    n_checked = 0
    for m in model.modules():
        st = getattr(m, "_dir_w16", None)
        if st is None:
            continue
        w = m.weight.detach()
        cout, rs, cin = st.shape
        r = m.kernel_size[0]
        w16 = torch.empty_like(st.w16)
        rot = None if st.w16_rot is None else torch.empty_like(st.w16_rot)
        L.check(L.lib().dir_conv_prep_weights_ex(L.ptr(w), cout, r, rs // r, cin, L.ptr(w16), L.ptr(rot), st.rot_mode, L.stream_ptr(w.device)), "prep")
        assert torch.equal(w16.view(torch.int16), st.w16.view(torch.int16))
        if rot is not None:
            assert torch.equal(rot.view(torch.int16), st.w16_rot.view(torch.int16))
        assert st.key == C._weight_key(m.weight)                  # no re-preparation at the next use
        n_checked += 1
    assert n_checked >= 52


def test_adam_falls_back_to_torch_for_what_the_kernel_does_not_take():
    from dirhip.optim import Adam
    p = torch.nn.Parameter(torch.randn(5, 3))
    p.grad = torch.randn(5, 3)
    q = p.detach().clone().requires_grad_(True)
    q.grad = p.grad.clone()
    a, b = Adam([p], lr=1e-2), torch.optim.Adam([q], lr=1e-2)     # CPU parameters: torch's own step
    a.step(); b.step()
    assert torch.equal(p, q)
    pg = torch.nn.Parameter(torch.randn(64, device="cuda"))
    pg.grad = torch.randn(64, device="cuda")
    qg = pg.detach().clone().requires_grad_(True)
    qg.grad = pg.grad.clone()
    a, b = Adam([pg], lr=1e-2, amsgrad=True), torch.optim.Adam([qg], lr=1e-2, amsgrad=True, foreach=False)
    a.step(); b.step()
    assert torch.equal(pg, qg)


def test_training_step_launches_no_library_optimizer_kernel():
    from torch.profiler import ProfilerActivity, profile
    from dirhip.loss import weighted_l1_loss
    from dirhip.optim import Adam
    from dirhip.train_loop import train_step
    model, eng, x = _model_with_grads()
    opt = Adam(eng.parameters(), lr=1e-3)
    y = torch.full((8, 1), 30.0, device="cuda")
    w = torch.ones(8, 1, device="cuda")
    for _ in range(2):
        train_step(eng, opt, x, y, w, 0, weighted_l1_loss, fds=False)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        train_step(eng, opt, x, y, w, 0, weighted_l1_loss, fds=False)
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert any("adam_step_kernel" in n for n in names)
    assert not any("FusedAdam" in n or "multi_tensor" in n or "conv_prep_weights_batched" in n for n in names), [n for n in names if "Adam" in n or "prep" in n]
    # gradients live in the engine's persistent slots: the optimizer's table is built once
    assert len(opt._tables) == 1
    key0 = opt._tables[0][0]
    train_step(eng, opt, x, y, w, 0, weighted_l1_loss, fds=False)
    assert opt._tables[0][0] == key0


def test_adam_loads_reference_layout_state_into_channels_last_model():
    """a reference optimizer checkpoint holds NCHW-contiguous moments; load_state_dict keeps those strides next to
    channels_last parameters. The one-linear-index kernel must not see them as they are: the moments are re-laid out once
    (same values) and the cached address table is rebuilt. Checked against torch.optim.Adam given the same state."""
    from dirhip.optim import Adam
    torch.manual_seed(0)
    shapes = [(64, 64, 3, 3), (128, 64, 1, 1), (64,), (8, 16)]
    def make():
        ps = []
        for i, s in enumerate(shapes):
            t = torch.randn(s, generator=torch.Generator().manual_seed(10 + i)).cuda()
            if t.dim() == 4:
                t = t.contiguous(memory_format=torch.channels_last)
            ps.append(torch.nn.Parameter(t))
        return ps
    ours_p, ref_p = make(), make()
    ours, ref = Adam(ours_p, lr=1e-2), torch.optim.Adam(ref_p, lr=1e-2, foreach=False, fused=False)
    def grads(seed):
        for i, (a, b) in enumerate(zip(ours_p, ref_p)):
            g = torch.randn(a.shape, generator=torch.Generator().manual_seed(seed + i)).cuda()
            if g.dim() == 4:
                g = g.contiguous(memory_format=torch.channels_last)
            a.grad, b.grad = g.clone(memory_format=torch.preserve_format), g.clone(memory_format=torch.preserve_format)
    grads(100)
    ours.step(); ref.step()
    # a reference-layout checkpoint: every moment NCHW-contiguous
    sd = ref.state_dict()
    for st in sd["state"].values():
        for k in ("exp_avg", "exp_avg_sq"):
            st[k] = st[k].contiguous().clone()
    tables_before = dict(ours._tables)
    ours.load_state_dict(copy.deepcopy(sd)); ref.load_state_dict(copy.deepcopy(sd))
    assert ours._tables == {} and tables_before                  # stale addresses dropped
    assert ours.state[ours_p[0]]["exp_avg"].stride() != ours_p[0].stride()          # torch kept the loaded strides
    for seed in (200, 300):
        grads(seed)
        ours.step(); ref.step()
    torch.cuda.synchronize()
    for a, b in zip(ours_p, ref_p):
        assert ours.state[a]["exp_avg"].stride() == a.stride()
        # (different gradients every step: exp_avg cancels, so one float32 rounding — torch's kernels contract a + w * (b - a) into an
        # fma, ours does not — is an ABSOLUTE 1e-8, not a relative one; a permuted moment would be off by O(0.1))
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-6)
        assert torch.allclose(ours.state[a]["exp_avg"], ref.state[b]["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(ours.state[a]["exp_avg_sq"], ref.state[b]["exp_avg_sq"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("momentum,weight_decay,nesterov,dampening", [(0.9, 1e-4, False, 0.0), (0.0, 0.0, False, 0.0), (0.9, 0.0, True, 0.0), (0.8, 1e-3, False, 0.1)])
def test_sgd_matches_torch_single_tensor_sgd_and_rewrites_conv_operands(momentum, weight_decay, nesterov, dampening):
    """dirhip.optim.SGD (--optimizer sgd of train.py:163-164) = ONE dir_sgd_step launch for all parameters: parameters and momentum buffers
    against torch.optim.SGD(foreach=False) over three steps with fresh gradients each step, the state_dict layout, and the bf16 conv
    operands the kernel rewrote (must equal dir_conv_prep_weights of the updated master weights, valid for the next forward)."""
    from dirhip import _lib as L
    from dirhip import conv as C
    from dirhip.optim import SGD
    model, eng, x = _model_with_grads()
    ref_params = [p.detach().clone().requires_grad_(True) for p in model.parameters()]
    kw = dict(lr=1e-2, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, dampening=dampening)
    ours = SGD(model.parameters(), **kw)
    theirs = torch.optim.SGD(ref_params, foreach=False, fused=False, **kw)
    gen = torch.Generator(device="cuda").manual_seed(9)
    for step in range(3):
        for rp, p in zip(ref_params, model.parameters()):
            g = torch.randn(p.shape, device="cuda", generator=gen) * 0.1
            if p.dim() == 4:
                g = g.contiguous(memory_format=torch.channels_last)
            p.grad.copy_(g)                                       # (the engine's persistent bucket slot)
            rp.grad = g.clone(memory_format=torch.preserve_format)
        theirs.step()
        ours.step()
    torch.cuda.synchronize()
    worst = 0.0
    for (n, p), rp in zip(model.named_parameters(), ref_params):
        scale = max(rp.detach().abs().max().item(), 1e-12)
        worst = max(worst, (p.detach().double() - rp.detach().double()).abs().max().item() / scale)
        if momentum:
            a, b = ours.state[p]["momentum_buffer"], theirs.state[rp]["momentum_buffer"]
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), n
        else:
            assert "momentum_buffer" not in ours.state[p] or ours.state[p]["momentum_buffer"] is None
    assert worst <= 1e-6, worst
    sd_o, sd_t = ours.state_dict(), theirs.state_dict()
    assert sd_o["param_groups"][0].keys() == sd_t["param_groups"][0].keys()
    n_checked = 0
    for m in model.modules():
        st = getattr(m, "_dir_w16", None)
        if st is None:
            continue
        w = m.weight.detach()
        cout, rs, cin = st.shape
        r = m.kernel_size[0]
        w16 = torch.empty_like(st.w16)
        rot = None if st.w16_rot is None else torch.empty_like(st.w16_rot)
        L.check(L.lib().dir_conv_prep_weights_ex(L.ptr(w), cout, r, rs // r, cin, L.ptr(w16), L.ptr(rot), st.rot_mode, L.stream_ptr(w.device)), "prep")
        assert torch.equal(w16.view(torch.int16), st.w16.view(torch.int16))
        if rot is not None:
            assert torch.equal(rot.view(torch.int16), st.w16_rot.view(torch.int16))
        assert st.key == C._weight_key(m.weight)
        n_checked += 1
    assert n_checked >= 52
    assert len(ours._tables) == 1


def test_sgd_train_step_launches_no_library_optimizer_kernel():
    from torch.profiler import ProfilerActivity, profile
    from dirhip.loss import weighted_l1_loss
    from dirhip.optim import SGD
    from dirhip.train_loop import train_step
    model, eng, x = _model_with_grads()
    opt = SGD(eng.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    y = torch.full((8, 1), 30.0, device="cuda")
    w = torch.ones(8, 1, device="cuda")
    for _ in range(2):
        train_step(eng, opt, x, y, w, 0, weighted_l1_loss, fds=False)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        train_step(eng, opt, x, y, w, 0, weighted_l1_loss, fds=False)
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert any("sgd_step_kernel" in n for n in names)
    assert not any("multi_tensor" in n or "conv_prep_weights_batched" in n or "FusedSgd" in n for n in names), [n for n in names if "multi_tensor" in n or "prep" in n][:5]
    assert len(opt._tables) == 1
