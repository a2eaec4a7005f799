"""-m gpu: the precision schedule of `train.py --amp_switch_epoch`: ONE engine / graph / optimizer whose conv stack runs on the
exact-float32 MFMA tile kernels first and on the bf16 MFMA kernels afterwards. The step after the switch must be exactly the step a bf16 engine
takes from the same weights and optimizer state (stale bf16 weight operands, a stale optimizer table or a float32 BatchNorm path would show)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    from dirhip.resnet import resnet50
    return resnet50(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9).cuda()


def test_switch_from_float32_to_bf16_equals_a_bf16_engine_started_from_the_same_state():
    from dirhip.optim import Adam
    from dirhip.parallel import DataParallelEngine
    from dirhip.train_loop import resolve_loss, train_step
    from torch.profiler import ProfilerActivity, profile
    loss_fn = resolve_loss("l1")
    g = torch.Generator(device="cuda").manual_seed(5)
    xs = [torch.randn(8, 3, 224, 224, device="cuda", generator=g) for _ in range(4)]
    ys = [torch.randint(20, 60, (8, 1), device="cuda", generator=g).float() for _ in range(4)]
    w = torch.ones(8, 1, device="cuda")
    torch.manual_seed(3)
    eng = DataParallelEngine(_model(), amp_dtype=None, channels_last=True)
    eng.train()
    opt = Adam(eng.parameters(), lr=1e-3)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for i in range(2):
            train_step(eng, opt, xs[i], ys[i], w, 0, loss_fn)
        torch.cuda.synchronize()
    names = {e.key for e in prof.key_averages()}
    assert any("conv_f32_tile_kernel" in n for n in names) and not any("conv_igemm" in n for n in names), sorted(names)[:10]
    snap_model = copy.deepcopy(eng.state_dict())
    snap_opt = copy.deepcopy(opt.state_dict())
    eng.set_amp_dtype(torch.bfloat16)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        la = [train_step(eng, opt, xs[i], ys[i], w, 0, loss_fn).item() for i in (2, 3)]
        torch.cuda.synchronize()
    names = {e.key for e in prof.key_averages()}
    assert any("conv_igemm" in n for n in names) and not any("conv_f32_tile_kernel<128, 128, 0," in n for n in names), sorted(names)[:10]
    # ---- the same two steps on a bf16 engine that starts from the snapshot
    torch.manual_seed(99)
    eng_b = DataParallelEngine(_model(), amp_dtype=torch.bfloat16, channels_last=True)
    eng_b.load_state_dict(snap_model)
    eng_b.train()
    opt_b = Adam(eng_b.parameters(), lr=1e-3)
    opt_b.load_state_dict(snap_opt)
    lb = [train_step(eng_b, opt_b, xs[i], ys[i], w, 0, loss_fn).item() for i in (2, 3)]
    assert la == lb, (la, lb)
    for (n, a), (_, b) in zip(eng.state_dict().items(), eng_b.state_dict().items()):
        assert torch.equal(a, b), n
    assert all(np.isfinite(la))
    # ... and back: float32 again on the same engine (the operands of the float32 kernels are the master weights themselves)
    eng.set_amp_dtype(None)
    l32 = train_step(eng, opt, xs[0], ys[0], w, 0, loss_fn).item()
    assert np.isfinite(l32)
