"""-m gpu: parity of the HIP path (through the C-ABI of libdir_hip.so) against the oracle and the golden
vectors of the reference, plus size-independent properties at BASELINE.json's full sizes.

Bars (north_star): bin indices / LDS weights bit-exact; FDS statistics and loss within 1e-5 relative
(written as assert_close(rtol=1e-5) below); where the HIP kernel and the numpy oracle evaluate the same
float32 expression with IEEE ops (calibration, bin smoothing, multiplier table) the test asks for
bit-equality, which is stricter.
"""
import json

import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import fds_oracle, loss_oracle

pytestmark = pytest.mark.gpu
BUFFERS = fds_oracle.FDSOracle.BUFFERS


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def make_fds(kw):
    from dirhip.fds import FDS
    return FDS(**kw).cuda()


def load_tables(F, g, prefix):
    for k in BUFFERS:
        getattr(F, k).copy_(dev(np.ascontiguousarray(g[prefix + k])))
    F._invalidate()


def test_library_loaded_is_in_tree():
    from dirhip import _lib
    assert _lib.lib().dir_abi_version() == _lib.ABI_VERSION == 4
    assert "imbalanced-regression_amd/dirhip/libdir_hip.so" in _lib.LIB_PATH


def test_bin_index_bit_exact(golden):
    from dirhip import ops
    g = golden("bin_index.npz")
    for i in range(int(g["n"])):
        start, num = (int(v) for v in g[f"params_{i}"])
        labels = dev(g[f"in_labels_{i}"][:, 0].astype(np.float32))
        bins, flags = ops.bin_index(labels, start, num)
        assert np.array_equal(bins.cpu().numpy(), g[f"ref_bins_{i}"]), i
        # fused path used by smooth(): same bins
        nb = num - start
        x = torch.zeros(labels.numel(), 8, device="cuda")
        m1 = torch.zeros(nb, 8, device="cuda")
        scale = torch.ones(nb, 8, device="cuda")
        bins2 = ops.smooth_fwd_(x, labels, start, num, m1, scale, m1)
        assert np.array_equal(bins2[:-1].cpu().numpy(), g[f"ref_bins_{i}"]), i
    rng = np.random.default_rng(0)
    for n in (1, 63, 64, 65, 1000, 5000, 191509):
        start, num = int(rng.integers(0, 5)), int(rng.integers(10, 120))
        lab = rng.integers(-2, num + 6, n).astype(np.float32)
        if n > 2000:
            lab[lab == start] = start + 1          # boundary label absent on one side (A.3)
        bins, flags = ops.bin_index(dev(lab), start, num)
        assert np.array_equal(bins.cpu().numpy(), fds_oracle.bin_index(lab, start, num)), n
        f = int(flags.item())
        assert bool(f & 1) == bool((lab == start).any()) and bool(f & 2) == bool((lab == num - 1).any())
        assert not (f & 12)


def test_label_flags_noninteger_and_nan():
    from dirhip import ops
    _, f = ops.bin_index(dev(np.array([3.0, 4.5, 7.0], np.float32)), 0, 10)
    assert int(f.item()) & 4
    _, f = ops.bin_index(dev(np.array([3.0, np.nan], np.float32)), 0, 10)
    assert int(f.item()) & 8
    _, f = ops.bin_index(dev(np.array([11.5, -0.5, 3.0], np.float32)), 0, 10)   # out-of-range fractions are fine
    assert not int(f.item()) & 12


@pytest.mark.parametrize("name", ["imdb", "agedb", "absent", "nomomentum", "frac", "fracnomom"])   # (frac*: non-integer labels, SURVEY A.8: gen_golden_r6.py)
def test_fds_state_machine_vs_reference_golden(golden, name):
    """Every buffer after every call of the epoch loop vs the reference's own outputs (1e-5 relative),
    and smooth() forward/backward vs the reference with its tables injected."""
    g = golden(f"fds_trace_{name}.npz")
    kw = json.loads(str(g["kw"]))
    F = make_fds(kw)
    O = fds_oracle.FDSOracle(**kw)
    for epoch in range(5):
        # -- smooth with the reference's tables injected
        P = make_fds(kw)
        load_tables(P, g, f"e{epoch}_pre_")
        x = dev(g[f"e{epoch}_in_x"]).requires_grad_(True)
        xin = x.clone()
        y = P.smooth(xin, dev(g[f"e{epoch}_in_labels_b"]), epoch)
        assert y.data_ptr() == xin.data_ptr()                        # in place, same tensor (A.2)
        y.backward(dev(g[f"e{epoch}_in_gy"]))
        assert_close(y.detach().cpu().numpy(), g[f"e{epoch}_ref_smooth"], rtol=2e-7, atol_scale=2e-7, msg=f"smooth e{epoch}")
        assert_close(x.grad.cpu().numpy(), g[f"e{epoch}_ref_gx"], rtol=2e-7, atol_scale=2e-7, msg=f"grad e{epoch}")
        # ... and bit-equal to the oracle fed the same tables (both IEEE float32, same op order)
        PO = fds_oracle.FDSOracle(**kw)
        for k in BUFFERS:
            setattr(PO, k, g[f"e{epoch}_pre_{k}"].copy())
        yo = PO.smooth(g[f"e{epoch}_in_x"].copy(), g[f"e{epoch}_in_labels_b"], epoch)
        assert np.array_equal(y.detach().cpu().numpy(), yo), f"smooth vs oracle e{epoch}"
        go = PO.smooth_grad(g[f"e{epoch}_in_gy"], g[f"e{epoch}_in_labels_b"], epoch)
        assert np.array_equal(x.grad.cpu().numpy(), go), f"grad vs oracle e{epoch}"
        # -- smooth with the HIP module's own evolved tables
        y2 = F.smooth(dev(g[f"e{epoch}_in_x"]), dev(g[f"e{epoch}_in_labels_b"]), epoch)
        assert_close(y2.cpu().numpy(), g[f"e{epoch}_ref_smooth"], msg=f"smooth(own tables) e{epoch}")
        # -- epoch tail (train.py:280-281)
        F.update_last_epoch_stats(epoch)
        O.update_last_epoch_stats(epoch)
        for k in BUFFERS:
            assert_close(getattr(F, k).cpu().numpy(), g[f"e{epoch}_mid_{k}"], msg=f"mid e{epoch} {k}")
        F.update_running_stats(dev(g[f"e{epoch}_in_feats"]), dev(g[f"e{epoch}_in_labels"]), epoch)
        O.update_running_stats(g[f"e{epoch}_in_feats"], g[f"e{epoch}_in_labels"], epoch)
        for k in BUFFERS:
            got = getattr(F, k).cpu().numpy()
            assert_close(got, g[f"e{epoch}_post_{k}"], msg=f"post e{epoch} {k} vs reference")
            assert_close(got, getattr(O, k), msg=f"post e{epoch} {k} vs oracle")
        assert (F.running_mean_last_epoch is F.running_mean) == bool(g[f"e{epoch}_alias"])          # A.1
        rv = F.running_var.cpu().numpy()
        assert np.array_equal(rv == 0, g[f"e{epoch}_post_running_var"] == 0)                        # A.9 exact zeros
        assert np.array_equal(F.num_samples_tracked.cpu().numpy(), g[f"e{epoch}_post_num_samples_tracked"])


def test_smooth_bins_bit_equal_oracle():
    from dirhip import ops
    rng = np.random.default_rng(1)
    for nb, c, (k, ks, s) in [(100, 2048, ("gaussian", 5, 2)), (97, 2048, ("gaussian", 9, 1)), (7, 33, ("triang", 5, 1)),
                              (3, 5, ("laplace", 3, 1.0)), (50, 12000, ("gaussian", 5, 2))]:
        O = fds_oracle.FDSOracle(c, bucket_num=nb, bucket_start=0, kernel=k, ks=ks, sigma=s)
        m = rng.normal(0, 1, (nb, c)).astype(np.float32)
        v = rng.uniform(0, 2, (nb, c)).astype(np.float32)
        sm, sv = ops.smooth_bins(dev(m), dev(v), dev(O.kernel_window))
        assert np.array_equal(sm.cpu().numpy(), O.smooth_bins(m))
        assert np.array_equal(sv.cpu().numpy(), O.smooth_bins(v))


def test_prepare_scale_bit_equal_oracle():
    from dirhip import ops
    rng = np.random.default_rng(2)
    nb, c = 100, 2048
    v1 = rng.uniform(0.001, 2, (nb, c)).astype(np.float32)
    v2 = rng.uniform(0.001, 2, (nb, c)).astype(np.float32)
    v1[3] = 0                      # whole row: sum < 1e-10
    v1[4] = 1e-15
    v1[5, ::7] = 0                 # some zero columns
    v2[6, :5] = [1e-9, 1e9, -1.0, 0.0, np.nan]
    for lo, hi in ((0.1, 10.0), (0.5, 2.0), (0.2, 5.0)):
        s = ops.prepare_scale(dev(v1), dev(v2), lo, hi).cpu().numpy()
        want = np.stack([fds_oracle.calibrate_scale(v1[b], v2[b], lo, hi) for b in range(nb)])
        assert np.array_equal(s, want, equal_nan=True)


def test_calibrate_mean_var_function(golden):
    from dirhip.utils import calibrate_mean_var
    g = golden("calibrate.npz")
    for i in range(int(g["n"])):
        lo, hi = (float(v) for v in g[f"clip_{i}"])
        x = dev(g[f"in_x_{i}"]).requires_grad_(True)
        xin = x.clone()                                            # (a non-leaf: two of the reference's three branches return / modify the input object)
        y = calibrate_mean_var(xin, dev(g[f"in_m1_{i}"]), dev(g[f"in_v1_{i}"]), dev(g[f"in_m2_{i}"]), dev(g[f"in_v2_{i}"]), lo, hi)
        v1 = g[f"in_v1_{i}"]
        if np.sum(v1, dtype=np.float32) < 1e-10 or (v1 == 0).any():
            assert y is xin, f"case {i}: utils.py:98-104 return the input object (calibrated in place where some v1 == 0)"
        else:
            assert y is not xin and y.data_ptr() != xin.data_ptr() and np.array_equal(xin.detach().cpu().numpy(), g[f"in_x_{i}"]), f"case {i}: utils.py:106-107 is out of place"
        assert_close(y.detach().cpu().numpy(), g[f"ref_y_{i}"], rtol=2e-7, atol_scale=2e-7, msg=f"case {i}")
        yo = fds_oracle.calibrate_mean_var(g[f"in_x_{i}"].copy(), g[f"in_m1_{i}"], g[f"in_v1_{i}"], g[f"in_m2_{i}"], g[f"in_v2_{i}"], lo, hi)
        assert np.array_equal(y.detach().cpu().numpy(), yo)
        y.sum().backward()
        s = fds_oracle.calibrate_scale(g[f"in_v1_{i}"], g[f"in_v2_{i}"], lo, hi)
        assert np.array_equal(x.grad.cpu().numpy(), np.broadcast_to(np.where(s < 0, 1, s), x.shape).astype(np.float32))


def test_scatter_stats_vs_float64_numpy():
    from dirhip import ops
    rng = np.random.default_rng(3)
    for n, c, nb in [(1, 8, 4), (130, 7, 5), (5000, 2048, 100), (20000, 130, 60), (3000, 12000, 50)]:
        bins = rng.integers(-1, nb, n).astype(np.int32)
        bins[bins == 2] = 3                                   # an empty bin
        feats = (rng.normal(0.5, 0.3, (n, c)) * (1 + bins[:, None] % 3)).astype(np.float32)
        feats[:, 1 % c] = 0.37109375                         # constant column -> m2 == 0 exactly
        cnt, mean, m2 = (t.cpu().numpy() for t in ops.scatter_stats(dev(feats), dev(bins), nb))
        for b in range(nb):
            rows = feats[bins == b].astype(np.float64)
            assert cnt[b] == rows.shape[0]
            if rows.shape[0] == 0:
                assert not mean[b].any() and not m2[b].any()
                continue
            assert_close(mean[b], rows.mean(0), rtol=1e-12, atol_scale=1e-13, msg=f"mean n={n} b={b}")
            want_m2 = ((rows - rows.mean(0)) ** 2).sum(0)
            assert_close(m2[b], want_m2, rtol=1e-9, atol_scale=1e-12, msg=f"m2 n={n} b={b}")
            assert m2[b, 1 % c] == 0.0 and mean[b, 1 % c] == 0.37109375


def test_losses_vs_reference_golden_and_oracle(golden):
    from dirhip import loss as hl
    g = golden("losses.npz")
    variants = [json.loads(str(v)) for v in g["variants"]]
    for b in (1, 8, 256, 1000):
        xs, ys, ws = g[f"in_x_{b}"], g[f"in_y_{b}"], g[f"in_w_{b}"]
        for vi, (kind, extra) in enumerate(variants):
            for use_w in (0, 1):
                x = dev(xs).requires_grad_(True)
                fn = getattr(hl, f"weighted_{kind}_loss")
                out = fn(x, dev(ys), dev(ws) if use_w else None, **extra)
                assert out.dim() == 0 and out.dtype == torch.float32
                out.backward()
                ref_l, ref_g = g[f"ref_loss_{b}_{vi}_{use_w}"], g[f"ref_grad_{b}_{vi}_{use_w}"]
                assert_close(out.item(), ref_l, msg=f"loss {kind} {extra} b={b} w={use_w}")
                assert_close(x.grad.cpu().numpy(), ref_g, msg=f"grad {kind} {extra} b={b} w={use_w}")
                lo, go = loss_oracle.weighted_loss(kind, xs, ys, ws if use_w else None, **extra)
                assert_close(out.item(), lo, msg="vs oracle")
                assert_close(x.grad.cpu().numpy(), go, msg="grad vs oracle")


def test_loss_large_n_and_upstream_gradient():
    """Dense-target size (NYUD2 'next' row): multi-workgroup path; non-unit upstream gradient."""
    from dirhip import loss as hl
    rng = np.random.default_rng(4)
    n = 32 * 114 * 152
    xs = rng.normal(3, 2, (n, 1)).astype(np.float32)
    ys = rng.uniform(0.7, 10, (n, 1)).astype(np.float32)
    ws = rng.uniform(0.1, 5, (n, 1)).astype(np.float32)
    for kind in ("mse", "l1", "huber", "focal_l1"):
        x = dev(xs).requires_grad_(True)
        out = getattr(hl, f"weighted_{kind}_loss")(x, dev(ys), dev(ws))
        (out * 3.0).backward()
        lo, go = loss_oracle.weighted_loss(kind, xs, ys, ws)
        assert_close(out.item(), lo, msg=kind)
        assert_close(x.grad.cpu().numpy(), 3.0 * go.astype(np.float64), msg=kind + " grad")


# ---------------------------------------------------------------------------------------------------
# Full-size (BASELINE config 2) properties: B=256, C=2048, Nb=100, N=191 509 — no oracle in the loop
# ---------------------------------------------------------------------------------------------------
def _long_tail_labels(rng, n):
    return np.clip(np.round(np.abs(rng.normal(0, 18, n)) + 20), 0, 120).astype(np.float32)


def test_full_size_whiten_recolor_round_trip():
    """encode -> calibrate -> re-measure: features whose per-bin statistics are (m1, v1), calibrated towards
    (m2, v2) with v2/v1 inside the clip range, must measure (m2, v2) — K1, K2, K5a, K5 at full size."""
    from dirhip import ops
    rng = np.random.default_rng(5)
    n, c, nb = 191509, 2048, 100
    labels = torch.as_tensor(_long_tail_labels(rng, n)).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    feats = torch.rand(n, c, device="cuda", generator=g) * 0.5 + 0.01 * labels[:, None]
    bins, _ = ops.bin_index(labels, 0, 100)
    cnt, m1, m2s = ops.scatter_stats(feats, bins, nb)
    denom = (cnt - 1).clamp(min=1)[:, None]
    v1 = (m2s / denom)
    tgt_m = torch.rand(nb, c, device="cuda", generator=g)
    tgt_v = v1 * (0.5 + 1.5 * torch.rand(nb, c, device="cuda", generator=g))       # ratio in [0.5, 2]
    scale = ops.prepare_scale(v1.float(), tgt_v.float(), 0.1, 10.0)
    out = feats.clone()
    ops.calibrate_fwd_(out, bins, m1.float(), scale, tgt_m.float())
    cnt2, mean2, m22 = ops.scatter_stats(out, bins, nb)
    assert torch.equal(cnt, cnt2)
    big = cnt >= 50                                                               # enough rows for a tight estimate
    assert big.sum() > 40
    v2 = m22 / denom
    # mean is reproduced to float32 round-off; variance to the float32 error of scale^2
    assert_close(mean2[big].cpu().numpy(), tgt_m[big].double().cpu().numpy(), rtol=1e-5, atol_scale=1e-5)
    assert_close(v2[big].cpu().numpy(), tgt_v[big].float().double().cpu().numpy(), rtol=1e-5, atol_scale=1e-5)
    # rows outside every bin (label > 99 without a '99' boundary row would be -1) are bit-untouched
    skipped = bins < 0
    assert torch.equal(out[skipped], feats[skipped])


def test_full_size_stats_permutation_and_merge_invariance():
    """K2 is a sum: a row permutation and a split + Chan merge must give the same (count, mean, M2);
    two runs on the same input must be bit-identical (fixed combination order, no atomics)."""
    from dirhip import ops
    rng = np.random.default_rng(6)
    n, c, nb = 191509, 2048, 100
    labels = torch.as_tensor(_long_tail_labels(rng, n)).cuda()
    g = torch.Generator(device="cuda").manual_seed(6)
    feats = torch.randn(n, c, device="cuda", generator=g).abs_() * 0.5 + 0.01 * labels[:, None]
    bins, _ = ops.bin_index(labels, 0, 100)
    a = ops.scatter_stats(feats, bins, nb)
    b = ops.scatter_stats(feats, bins, nb)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    perm = torch.randperm(n, device="cuda", generator=g)
    p = ops.scatter_stats(feats[perm].contiguous(), bins[perm].contiguous(), nb)
    assert torch.equal(a[0], p[0])
    assert_close(p[1].cpu().numpy(), a[1].cpu().numpy(), rtol=1e-12, atol_scale=1e-13)
    assert_close(p[2].cpu().numpy(), a[2].cpu().numpy(), rtol=1e-9, atol_scale=1e-12)
    half = n // 2
    s1 = ops.scatter_stats(feats[:half].contiguous(), bins[:half].contiguous(), nb)
    s2 = ops.scatter_stats(feats[half:].contiguous(), bins[half:].contiguous(), nb)
    tot = s1[0] + s2[0]
    safe = tot.clamp(min=1)[:, None]
    mean = (s1[1] * s1[0][:, None] + s2[1] * s2[0][:, None]) / safe
    m2 = s1[2] + s1[0][:, None] * (s1[1] - mean) ** 2 + s2[2] + s2[0][:, None] * (s2[1] - mean) ** 2
    assert torch.equal(tot, a[0])
    assert_close(mean.cpu().numpy(), a[1].cpu().numpy(), rtol=1e-12, atol_scale=1e-13)
    assert_close(m2.cpu().numpy(), a[2].cpu().numpy(), rtol=1e-9, atol_scale=1e-12)


def test_full_size_smooth_identity_and_linearity():
    from dirhip import ops
    from dirhip.fds import FDS
    F = FDS(2048).cuda()
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(256, 2048, device="cuda", generator=g)
    labels = torch.randint(0, 121, (256, 1), device="cuda", generator=g).float()
    # fresh tables (mean 0 / var 1 -> factor 1, m1 = m2 = 0): epoch-1 smoothing is the identity (A.4)
    y = F.smooth(x.clone(), labels, 1)
    assert torch.equal(y, x)
    # bin smoothing is linear and preserves constants up to float32 round-off
    w = F._window_on(x.device)
    a = torch.randn(100, 2048, device="cuda", generator=g)
    b = torch.randn(100, 2048, device="cuda", generator=g)
    sa, sb = ops.smooth_bins(a, b, w)
    sab, _ = ops.smooth_bins(a + b, b, w)
    assert_close(sab.cpu().numpy(), (sa + sb).cpu().numpy(), rtol=1e-5, atol_scale=1e-6)
    ones, _ = ops.smooth_bins(torch.full((100, 2048), 3.0, device="cuda"), b, w)
    assert_close(ones.cpu().numpy(), np.full((100, 2048), 3.0), rtol=1e-6, atol_scale=1e-6)


def test_full_size_smooth_bit_equal_oracle():
    """BASELINE.json's training shape (B=256 rows x 2048 features, 100 bins): smooth() forward and backward bit-for-bit
    against the oracle (fds.py:115-144) when both hold the SAME tables. The tables come from the oracle's own two epochs
    of update_running_stats / update_last_epoch_stats over realistic features (post-ReLU, per-channel scale spread), so the
    calibration runs on non-trivial means / variances, with zero-variance channels (A.9 guard: v1 == 0 -> untouched),
    factors clipped at 10 (the 5-tap window keeps >= 0.25 of a bin's own variance, so 0.1 is unreachable with ks=5) and
    bins the batch never visits."""
    from dirhip.fds import FDS
    rng = np.random.default_rng(20260924)
    C, NB = 2048, 100
    O = fds_oracle.FDSOracle(C)
    chan_scale = np.exp(rng.normal(0.0, 1.0, C)).astype(np.float32)
    chan_scale[::97] = 0.0                                           # dead channels -> variance exactly 0
    for epoch in range(2):
        n = 6000
        labels = np.clip(np.round(rng.gamma(4.0, 8.0, n)), 0, 120).astype(np.float32)
        labels[labels > 95] = 95                                     # bins 96.. stay at their initial (0, 1) rows
        feats = np.maximum(rng.normal(0.2, 1.0, (n, C)).astype(np.float32) * chan_scale * (1 + labels[:, None] / 60), 0)
        feats[:, 5::211] *= np.where(labels[:, None] % 2 == 0, 30.0, 0.02).astype(np.float32)   # odd bins: factor clips at 10
        O.update_last_epoch_stats(epoch)
        O.update_running_stats(feats, labels, epoch)
    O.update_last_epoch_stats(2)
    F = FDS(C).cuda()
    for k in BUFFERS:
        getattr(F, k).copy_(dev(np.ascontiguousarray(getattr(O, k))))
    F._invalidate()
    B = 256
    lab = np.clip(np.round(rng.gamma(4.0, 8.0, (B, 1))), 0, 120).astype(np.float32)
    lab[:4, 0] = (0.0, 99.0, 120.0, 97.0)                            # edge bins, incl. the clamp at the top (utils / fds.py:119-123)
    x = np.maximum(rng.normal(0.2, 1.0, (B, C)), 0).astype(np.float32) * chan_scale
    gy = rng.normal(0, 1, (B, C)).astype(np.float32)
    s = np.stack([fds_oracle.calibrate_scale(O.running_var_last_epoch[b], O.smoothed_var_last_epoch[b]) for b in range(NB)])
    assert (s < 0).any() and (s == np.sqrt(np.float32(10.0))).any() and ((s > 0.5) & (s < 3.0)).any()
    xt = dev(x).requires_grad_(True)
    y = F.smooth(xt.clone(), dev(lab), 2)
    y.backward(dev(gy))
    yo = O.smooth(x.copy(), lab, 2)
    go = O.smooth_grad(gy, lab, 2)
    assert not np.array_equal(yo, x)
    assert np.array_equal(y.detach().cpu().numpy(), yo)
    assert np.array_equal(xt.grad.cpu().numpy(), go)


# ---------------------------------------------------------------------------------------------------
# STS-B FDS variant: histogram-edge buckets, clip [0.5, 2] with v1 <= 0 / v2 < 0 guards,
# empty-bucket fill — vs the reference's own outputs (golden) and the oracle
# ---------------------------------------------------------------------------------------------------
def test_stsb_fds_state_machine_vs_reference_golden(golden):
    from dirhip.fds_stsb import FDS as StsbFDS
    from oracle import fds_stsb_oracle as so
    g = golden("fds_trace_stsb.npz")
    kw = json.loads(str(g["kw"]))
    F = StsbFDS(**kw).cuda()
    O = so.FDSStsbOracle(**kw)
    for epoch in range(4):
        bins = F._bins(dev(g[f"e{epoch}_in_labels"])).cpu().numpy()
        assert np.array_equal(bins, g[f"e{epoch}_ref_buckets"] - kw["bucket_start"])           # bucket indices: bit-exact
        P = StsbFDS(**kw).cuda()
        load_tables(P, g, f"e{epoch}_pre_")
        x = dev(g[f"e{epoch}_in_x"]).requires_grad_(True)
        xin = x.clone()
        y = P.smooth(xin, dev(g[f"e{epoch}_in_labels_b"]), epoch)
        assert y.data_ptr() == xin.data_ptr()
        y.backward(dev(g[f"e{epoch}_in_gy"]))
        assert_close(y.detach().cpu().numpy(), g[f"e{epoch}_ref_smooth"], rtol=2e-7, atol_scale=2e-7, msg=f"smooth e{epoch}")
        assert_close(x.grad.cpu().numpy(), g[f"e{epoch}_ref_gx"], rtol=2e-7, atol_scale=2e-7, msg=f"grad e{epoch}")
        F.update_last_epoch_stats(epoch)
        O.update_last_epoch_stats(epoch)
        F.update_running_stats(dev(g[f"e{epoch}_in_feats"]), dev(g[f"e{epoch}_in_labels"]), epoch)
        O.update_running_stats(g[f"e{epoch}_in_feats"], g[f"e{epoch}_in_labels"], epoch)
        for k in BUFFERS:
            got = getattr(F, k).cpu().numpy()
            assert_close(got, g[f"e{epoch}_post_{k}"], msg=f"post e{epoch} {k} vs reference")
            assert_close(got, getattr(O, k), msg=f"post e{epoch} {k} vs oracle")
        assert np.array_equal(F.num_samples_tracked.cpu().numpy(), g[f"e{epoch}_post_num_samples_tracked"])


def test_stsb_calibrate_mean_var(golden):
    from dirhip.fds_stsb import calibrate_mean_var
    from oracle import fds_stsb_oracle as so
    g = golden("calibrate_stsb.npz")
    for i in range(int(g["n"])):
        lo, hi = (float(v) for v in g[f"clip_{i}"])
        y = calibrate_mean_var(dev(g[f"in_x_{i}"]), dev(g[f"in_m1_{i}"]), dev(g[f"in_v1_{i}"]), dev(g[f"in_m2_{i}"]), dev(g[f"in_v2_{i}"]), lo, hi)
        assert_close(y.cpu().numpy(), g[f"ref_y_{i}"], rtol=2e-7, atol_scale=2e-7, msg=f"case {i}")
        yo = so.calibrate_mean_var(g[f"in_x_{i}"].copy(), g[f"in_m1_{i}"], g[f"in_v1_{i}"], g[f"in_m2_{i}"], g[f"in_v2_{i}"], lo, hi)
        assert np.array_equal(y.cpu().numpy(), yo)


def test_stsb_full_size_shapes():
    """BASELINE configs[4] shapes: features [128, 12000], 50 buckets on [0, 5]: smooth is the identity on fresh tables,
    bucket indices equal the oracle's for random scores, statistics are bit-reproducible."""
    from dirhip.fds_stsb import FDS as StsbFDS
    from oracle import fds_stsb_oracle as so
    rng = np.random.default_rng(8)
    F = StsbFDS(12000).cuda()
    labels = np.round(rng.uniform(0, 5, 128), 3).astype(np.float32)
    labels[:3] = [0.0, 5.0, 2.5]
    x = torch.randn(128, 12000, device="cuda")
    assert torch.equal(F.smooth(x.clone(), dev(labels[:, None]), 1), x)
    assert np.array_equal(F._bins(dev(labels)).cpu().numpy(), so.bucket_idx(labels, 0, 50))
    feats = torch.randn(5249, 12000, device="cuda").abs_()
    lab = dev(np.round(rng.uniform(0, 5, 5249), 3).astype(np.float32))
    a = F.local_stats(feats, lab)
    b = F.local_stats(feats, lab)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    F.update_last_epoch_stats(0)
    F.update_running_stats(feats, lab, 0)
    assert float(F.num_samples_tracked.sum()) == 5249


def test_stsb_guard_modes_vs_oracle():
    """guard_mode 1 (reference as executed on torch >= 1.2: any guarded column -> whole row untouched) and guard_mode 2
    (torch 0.4.1 / intended: only the guarded columns untouched) against the oracle, bit for bit."""
    from dirhip import ops
    from oracle import fds_stsb_oracle as so
    rng = np.random.default_rng(12)
    nb, c = 50, 12000
    v1 = rng.uniform(0.01, 2, (nb, c)).astype(np.float32)
    v2 = rng.uniform(0.01, 2, (nb, c)).astype(np.float32)
    v1[3, ::11] = 0.0
    v1[4, 5] = -0.25
    v2[5, 7] = -1e-6
    v1[6] = 1e-15
    for mode, per_col in ((1, False), (2, True)):
        s = ops.prepare_scale(dev(v1), dev(v2), 0.5, 2.0, guard_mode=mode).cpu().numpy()
        want = np.stack([so.calibrate_scale(v1[b], v2[b], 0.5, 2.0, per_column_guard=per_col) for b in range(nb)])
        assert np.array_equal(s, want), mode


# ---------------------------------------------------------------------------------------------------
# NYUD2 dense FDS variant: per-pixel buckets, NCHW maps, clip [0.2, 5], not in place, snapshots
# ---------------------------------------------------------------------------------------------------
def test_nyud2_fds_state_machine_vs_reference_golden(golden):
    from dirhip.fds_nyud2 import FDS as DenseFDS
    from oracle import fds_nyud2_oracle as no
    g = golden("fds_trace_nyud2.npz")
    kw = json.loads(str(g["kw"]))
    F = DenseFDS(**kw).cuda()
    O = no.FDSNyud2Oracle(**kw)
    for epoch in range(4):
        bins = F._bins(dev(g[f"e{epoch}_in_labels"]).squeeze(1)).cpu().numpy()
        assert np.array_equal(bins, no.bucket_idx(g[f"e{epoch}_in_labels"].reshape(-1), kw["bucket_start"], kw["bucket_num"]) - kw["bucket_start"])
        P = DenseFDS(**kw).cuda()
        load_tables(P, g, f"e{epoch}_pre_")
        x = dev(g[f"e{epoch}_in_x"]).requires_grad_(True)
        xin = x * 1.0
        y = P.smooth(xin, dev(g[f"e{epoch}_in_labels_b"]), epoch)
        assert y.shape == x.shape
        assert torch.equal(xin.detach(), x.detach())                           # caller's tensor untouched (not in place)
        if epoch >= kw["start_smooth"]:
            y.backward(dev(g[f"e{epoch}_in_gy"]))
            assert_close(x.grad.cpu().numpy(), g[f"e{epoch}_ref_gx"], rtol=2e-7, atol_scale=2e-7, msg=f"grad e{epoch}")
        assert_close(y.detach().cpu().numpy(), g[f"e{epoch}_ref_smooth"], rtol=2e-7, atol_scale=2e-7, msg=f"smooth e{epoch}")
        F.update_last_epoch_stats(epoch)
        O.update_last_epoch_stats(epoch)
        F.update_running_stats(dev(g[f"e{epoch}_in_feats"]), dev(g[f"e{epoch}_in_labels"]), epoch)
        O.update_running_stats(g[f"e{epoch}_in_feats"], g[f"e{epoch}_in_labels"], epoch)
        for k in BUFFERS:
            got = getattr(F, k).cpu().numpy()
            assert_close(got, g[f"e{epoch}_post_{k}"], msg=f"post e{epoch} {k} vs reference")
            assert_close(got, getattr(O, k), msg=f"post e{epoch} {k} vs oracle")
        assert (F.running_mean_last_epoch is F.running_mean) == bool(g[f"e{epoch}_alias"])
        assert np.array_equal(F.num_samples_tracked.cpu().numpy(), g[f"e{epoch}_post_num_samples_tracked"])


def test_nyud2_full_size_whiten_recolor_and_speed():
    """BASELINE configs[3] shapes: features [32, 128, 114, 152] (284 MB), depth ~ U(0.7, 10): statistics -> calibrate
    towards random targets -> re-measure (K-scaled bins, narrow-row scatter and calibrate paths at full size)."""
    from dirhip import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    b, c, h, w = 32, 128, 114, 152
    depth = torch.rand(b, 1, h, w, device="cuda", generator=g) * 9.3 + 0.7
    feats = torch.rand(b, c, h, w, device="cuda", generator=g) * 0.5 + 0.05 * depth
    rows = feats.permute(0, 2, 3, 1).contiguous().view(-1, c)
    bins = ops.bin_scaled(depth.reshape(-1), 10.0, 7, 100)
    assert int(bins.min()) == 0 and int(bins.max()) == 92
    nb = 93
    cnt, m1, m2s = ops.scatter_stats(rows, bins, nb)
    assert float(cnt.sum()) == rows.shape[0]
    v1 = m2s / (cnt - 1).clamp(min=1)[:, None]
    tgt_m = torch.rand(nb, c, device="cuda", generator=g)
    tgt_v = v1 * (0.5 + 1.5 * torch.rand(nb, c, device="cuda", generator=g))
    scale = ops.prepare_scale(v1.float(), tgt_v.float(), 0.2, 5.0, guard_mode=1)
    out = rows.clone()
    ops.calibrate_fwd_(out, bins, m1.float(), scale, tgt_m.float())
    cnt2, mean2, m22 = ops.scatter_stats(out, bins, nb)
    assert torch.equal(cnt, cnt2)
    assert_close(mean2.cpu().numpy(), tgt_m.double().cpu().numpy(), rtol=1e-5, atol_scale=1e-5)
    assert_close((m22 / (cnt - 1).clamp(min=1)[:, None]).cpu().numpy(), tgt_v.float().double().cpu().numpy(), rtol=1e-5, atol_scale=1e-5)
    # gradient path: dx = dy * scale
    dy = torch.randn(rows.shape, device="cuda", generator=g)
    dx = ops.calibrate_bwd(dy, bins, scale)
    assert torch.equal(dx, dy * scale[bins.long()])


# ---------------------------------------------------------------------------------------------------
# Round 3: calibration with the bin-statistic tables staged in LDS (large batches) and on NCHW maps (NYUD2)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("b,c,nb", [(65536, 2048, 100), (5003, 2048, 97), (4096, 12000, 50), (100003, 128, 93), (7001, 520, 100), (6000, 2048, 300)])
def test_calibrate_lds_staged_bit_identical_to_row_kernel(b, c, nb):
    from dirhip import _lib as L
    """dir_fds_calibrate_fwd_lds (tables resident in LDS, rows streamed in natural order, 4 rows in flight per lane) against the
    per-row kernel dir_fds_calibrate_fwd: same arithmetic -> identical bits, rows with bin < 0 untouched, ragged row counts,
    partial column tiles (C = 520, 12000), narrow rows (C = 128) and a table too tall for a 128-column tile (nb = 300 -> 32)."""
    from dirhip import ops
    g = torch.Generator(device="cuda").manual_seed(b + c)
    x = torch.randn(b, c, device="cuda", generator=g)
    bins = torch.randint(-1, nb, (b,), device="cuda", generator=g, dtype=torch.int32)
    m1 = torch.randn(nb, c, device="cuda", generator=g)
    sc = torch.rand(nb, c, device="cuda", generator=g) + 0.5
    sc[torch.rand(nb, c, device="cuda", generator=g) < 0.05] = -1.0          # "leave untouched" columns (v1 == 0, utils.py:100-104)
    m2 = torch.randn(nb, c, device="cuda", generator=g)
    want = x.clone()
    L.check(L.lib().dir_fds_calibrate_fwd(L.ptr(want), L.DIR_F32, L.ptr(bins), b, c, L.ptr(m1), L.ptr(sc), L.ptr(m2), L.stream_ptr(x.device)), "row kernel")
    got = ops.calibrate_fwd_lds_(x.clone(), bins, m1, sc, m2)
    assert torch.equal(got, want)
    assert torch.equal(got[bins < 0], x[bins < 0])


def test_smooth_large_batch_takes_lds_path_and_matches_oracle():
    """FDS.smooth at B = 8192 (dir_fds_smooth_fwd -> K1 + the LDS-staged calibration) bit-equal to the numpy oracle's
    calibrate_mean_var given the same tables and bins."""
    from dirhip import ops
    rng = np.random.default_rng(5)
    b, c, nb = 8192, 2048, 100
    x = rng.normal(0, 1, (b, c)).astype(np.float32)
    labels = np.clip(np.round(np.abs(rng.normal(0, 18, b)) + 20), 0, 120).astype(np.float32)
    m1 = rng.normal(0, 1, (nb, c)).astype(np.float32)
    v1 = rng.uniform(0.5, 2, (nb, c)).astype(np.float32)
    v2 = rng.uniform(0.5, 2, (nb, c)).astype(np.float32)
    m2 = rng.normal(0, 1, (nb, c)).astype(np.float32)
    scale = ops.prepare_scale(dev(v1), dev(v2), 0.1, 10.0)
    xd = dev(x.copy())
    bins = ops.smooth_fwd_(xd, dev(labels), 0, 100, dev(m1), scale, dev(m2))[:b].cpu().numpy()
    want = x.copy()
    for bb in np.unique(bins):
        if bb < 0:
            continue
        rows = bins == bb
        want[rows] = fds_oracle.calibrate_mean_var(x[rows], m1[bb], v1[bb], m2[bb], v2[bb])
    assert np.array_equal(xd.cpu().numpy(), want)


def test_nyud2_nchw_calibration_bit_identical_to_row_form():
    from dirhip import _lib as L
    """dir_fds_calibrate_{fwd,bwd}_nchw on the [32, 128, 114, 152] map (tables transposed in LDS, 4 pixels per lane) against the
    permute -> row kernel -> permute chain it replaces (nyud2-dir/models/fds.py:136,149): identical bits, input untouched."""
    from dirhip import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    b, c, h, w, nb = 32, 128, 114, 152, 93
    depth = torch.rand(b, 1, h, w, device="cuda", generator=g) * 9.3 + 0.7
    x = torch.rand(b, c, h, w, device="cuda", generator=g)
    bins = ops.bin_scaled(depth.reshape(-1), 10.0, 7, 100)
    bins[::1013] = -1
    m1, m2 = (torch.randn(nb, c, device="cuda", generator=g) for _ in range(2))
    sc = torch.rand(nb, c, device="cuda", generator=g) + 0.5
    sc[torch.rand(nb, c, device="cuda", generator=g) < 0.05] = -1.0
    keep = x.clone()
    y = ops.calibrate_nchw(x, bins, m1, sc, m2)
    assert y is not None and torch.equal(x, keep)
    rows = x.permute(0, 2, 3, 1).contiguous().view(-1, c)
    L.check(L.lib().dir_fds_calibrate_fwd(L.ptr(rows), L.DIR_F32, L.ptr(bins), rows.shape[0], c, L.ptr(m1), L.ptr(sc), L.ptr(m2), L.stream_ptr(x.device)), "row kernel")
    assert torch.equal(y, rows.view(b, h, w, c).permute(0, 3, 1, 2))
    dy = torch.randn(b, c, h, w, device="cuda", generator=g)
    dx = ops.calibrate_bwd_nchw(dy, bins, sc)
    want = ops.calibrate_bwd(dy.permute(0, 2, 3, 1).contiguous().view(-1, c), bins, sc).view(b, h, w, c).permute(0, 3, 1, 2)
    assert torch.equal(dx, want)
    # geometry the kernel does not take -> None (the module then falls back to the row form)
    assert ops.calibrate_nchw(torch.rand(2, 128, 5, 5, device="cuda"), torch.zeros(50, dtype=torch.int32, device="cuda"), m1, sc, m2) is None
