"""Pins oracle/ against the golden vectors produced by the upstream reference
(tests/golden/gen_golden.py). CPU only. Tolerances: bit-exact for windows, LDS weights,
bin indices, calibration given identical tables; 1e-5 relative (north_star) for FDS
statistics and losses — the measured gap is ~3e-7."""
import json

import numpy as np
import pytest

from conftest import assert_close, relerr
from oracle import fds_oracle, lds_oracle, loss_oracle

BUFFERS = fds_oracle.FDSOracle.BUFFERS


def test_windows_bit_exact(golden):
    g = golden("windows.npz")
    for i, spec in enumerate(g["grid"]):
        k, ks, s = str(spec).split(",")
        ks, s = int(ks), float(s)
        assert np.array_equal(fds_oracle.fds_kernel_window(k, ks, s), g[f"ref_fds_{i}"]), spec
        assert np.array_equal(np.asarray(lds_oracle.get_lds_kernel_window(k, ks, s)), g[f"ref_lds_{i}"]), spec


def test_windows_known_answers():
    # SURVEY.md Appendix C
    w = lds_oracle.get_lds_kernel_window("gaussian", 5, 2)
    assert w[0] == 0.8582852377730947 and w[1] == 0.9458276490853568 and w[2] == 1.0
    f = fds_oracle.fds_kernel_window("gaussian", 5, 2).astype(np.float64)
    assert f[0] == 0.1862506866455078 and f[2] == 0.2170032560825348


def test_lds_weights_bit_exact(golden):
    g = golden("lds_weights.npz")
    for sname in ("agedb", "synth", "frac", "tiny"):
        labels = g[f"in_labels_{sname}"]
        for ci, spec in enumerate(g["configs"]):
            rw, lds, k, ks, s = str(spec).split(",")
            w = lds_oracle.prepare_weights(labels, rw, lds=bool(int(lds)), lds_kernel=k, lds_ks=int(ks), lds_sigma=float(s))
            ref = g[f"ref_w_{sname}_{ci}"]
            if rw == "none":
                assert w is None and ref.size == 0
                continue
            assert w.dtype == np.float32
            assert np.array_equal(w, ref), (sname, spec)


def test_lds_agedb_known_answer(golden):
    # first five weights for ages [31,44,34,74,62], min/max
    g = golden("lds_weights.npz")
    w = lds_oracle.prepare_weights(g["in_labels_agedb"], "sqrt_inv", lds=True, lds_kernel="gaussian", lds_ks=5, lds_sigma=2)
    assert list(g["in_labels_agedb"][:5]) == [31, 44, 34, 74, 62]
    np.testing.assert_allclose(w[:5], [0.7677357197, 0.8096965551, 0.7403168678, 1.4011092186, 1.0261026621], rtol=1e-7)
    np.testing.assert_allclose([w.min(), w.max()], [0.7320062518, 22.0098342896], rtol=1e-7)


def test_bin_index_bit_exact(golden):
    g = golden("bin_index.npz")
    for i in range(int(g["n"])):
        start, num = g[f"params_{i}"]
        bins = fds_oracle.bin_index(g[f"in_labels_{i}"], int(start), int(num))
        assert np.array_equal(bins, g[f"ref_bins_{i}"]), i


def test_calibrate_bit_exact(golden):
    g = golden("calibrate.npz")
    for i in range(int(g["n"])):
        lo, hi = g[f"clip_{i}"]
        y = fds_oracle.calibrate_mean_var(g[f"in_x_{i}"].copy(), g[f"in_m1_{i}"], g[f"in_v1_{i}"],
                                          g[f"in_m2_{i}"], g[f"in_v2_{i}"], lo, hi)
        assert np.array_equal(y, g[f"ref_y_{i}"]), i


@pytest.mark.parametrize("name", ["imdb", "agedb", "absent", "nomomentum", "frac", "fracnomom"])    # (frac*: non-integer labels; gen_golden_r6.py)
def test_fds_state_machine(golden, name):
    g = golden(f"fds_trace_{name}.npz")
    kw = json.loads(str(g["kw"]))
    O = fds_oracle.FDSOracle(**kw)
    for epoch in range(5):
        # (1) smooth with the oracle's own evolved tables: 1e-5 relative
        y = O.smooth(g[f"e{epoch}_in_x"].copy(), g[f"e{epoch}_in_labels_b"], epoch)
        # (x-m1)*s+m2 cancels, so element errors are measured against the operand scale (1e-2*max|y|)
        assert_close(y, g[f"e{epoch}_ref_smooth"], msg=f"smooth e{epoch}")
        gx = O.smooth_grad(g[f"e{epoch}_in_gy"], g[f"e{epoch}_in_labels_b"], epoch)
        assert_close(gx, g[f"e{epoch}_ref_gx"], msg=f"smooth grad e{epoch}")
        # (2) smooth with the reference's tables injected: equal up to the reference's own sqrt.
        # torch-CPU sqrt (MKL/AVX512 path) is NOT correctly rounded: 0.7 % of float32 inputs come out
        # 1 ulp off IEEE (measured in the build container), numpy / HIP sqrtf are exact -> <= ~1 ulp.
        P = fds_oracle.FDSOracle(**kw)
        for k in BUFFERS:
            setattr(P, k, g[f"e{epoch}_pre_{k}"].copy())
        y2 = P.smooth(g[f"e{epoch}_in_x"].copy(), g[f"e{epoch}_in_labels_b"], epoch)
        assert_close(y2, g[f"e{epoch}_ref_smooth"], rtol=2e-7, atol_scale=2e-7, msg=f"smooth(injected) e{epoch}")
        assert np.mean(y2 != g[f"e{epoch}_ref_smooth"]) < 0.02
        gx2 = P.smooth_grad(g[f"e{epoch}_in_gy"], g[f"e{epoch}_in_labels_b"], epoch)
        assert_close(gx2, g[f"e{epoch}_ref_gx"], rtol=2e-7, atol_scale=2e-7, msg=f"grad(injected) e{epoch}")
        # (3) epoch tail
        O.update_last_epoch_stats(epoch)
        for k in BUFFERS:
            assert_close(getattr(O, k), g[f"e{epoch}_mid_{k}"], msg=f"mid e{epoch} {k}")
        O.update_running_stats(g[f"e{epoch}_in_feats"], g[f"e{epoch}_in_labels"], epoch)
        for k in BUFFERS:
            assert_close(getattr(O, k), g[f"e{epoch}_post_{k}"], msg=f"post e{epoch} {k}")
        assert (O.running_mean_last_epoch is O.running_mean) == bool(g[f"e{epoch}_alias"])   # A.1
        # zero-variance columns must be exactly zero like the reference's (A.9)
        assert np.array_equal(O.running_var == 0, g[f"e{epoch}_post_running_var"] == 0)


def test_losses(golden):
    g = golden("losses.npz")
    variants = [json.loads(str(v)) for v in g["variants"]]
    for b in (1, 8, 256, 1000):
        x, y, w = g[f"in_x_{b}"], g[f"in_y_{b}"], g[f"in_w_{b}"]
        for vi, (kind, extra) in enumerate(variants):
            for use_w in (0, 1):
                loss, grad = loss_oracle.weighted_loss(kind, x, y, w if use_w else None, **extra)
                assert relerr(loss, g[f"ref_loss_{b}_{vi}_{use_w}"]) < 1e-5, (b, kind, extra, use_w)
                ref_g = g[f"ref_grad_{b}_{vi}_{use_w}"]
                assert relerr(grad, ref_g, floor=1e-3 * float(np.abs(ref_g).max())) < 1e-5, (b, kind, extra, use_w)


@pytest.mark.needs_reference
def test_oracle_vs_live_reference_random():
    """Extra pin when /root/reference is present: random FDS sequences vs the live reference."""
    import torch
    from oracle import refshim
    rng = np.random.default_rng(42)
    for trial in range(3):
        start, num, c, n = int(rng.integers(0, 4)), int(rng.integers(12, 60)), 16, 300
        kw = dict(feature_dim=c, bucket_num=num, bucket_start=start, start_update=0, start_smooth=1,
                  kernel="gaussian", ks=5, sigma=2, momentum=0.9)
        R = refshim.make_fds(**kw)
        O = fds_oracle.FDSOracle(**kw)
        for epoch in range(4):
            labels = np.clip(np.round(rng.normal(num / 2, num / 3, n)), 0, num + 4).astype(np.float32)
            feats = rng.normal(0.5, 0.3, (n, c)).astype(np.float32)
            with refshim.cuda_identity():
                R.update_last_epoch_stats(epoch)
                R.update_running_stats(torch.tensor(feats), torch.tensor(labels), epoch)
            O.update_last_epoch_stats(epoch)
            O.update_running_stats(feats, labels, epoch)
            for k in BUFFERS:
                assert_close(getattr(O, k), getattr(R, k).numpy(), msg=f"{trial} {epoch} {k}")


# ---- STS-B FDS variant -----------------------------------------------------------------
def test_stsb_oracle_state_machine(golden):
    from oracle import fds_stsb_oracle as so
    g = golden("fds_trace_stsb.npz")
    kw = json.loads(str(g["kw"]))
    O = so.FDSStsbOracle(**kw)
    for epoch in range(4):
        assert np.array_equal(so.bucket_idx(g[f"e{epoch}_in_labels"], kw["bucket_start"], kw["bucket_num"]), g[f"e{epoch}_ref_buckets"])
        y = O.smooth(g[f"e{epoch}_in_x"].copy(), g[f"e{epoch}_in_labels_b"], epoch)
        assert_close(y, g[f"e{epoch}_ref_smooth"], msg=f"smooth e{epoch}")
        assert_close(O.smooth_grad(g[f"e{epoch}_in_gy"], g[f"e{epoch}_in_labels_b"], epoch), g[f"e{epoch}_ref_gx"], msg=f"grad e{epoch}")
        O.update_last_epoch_stats(epoch)
        O.update_running_stats(g[f"e{epoch}_in_feats"], g[f"e{epoch}_in_labels"], epoch)
        for k in BUFFERS:
            assert_close(getattr(O, k), g[f"e{epoch}_post_{k}"], msg=f"post e{epoch} {k}")


def test_stsb_calibrate_oracle(golden):
    from oracle import fds_stsb_oracle as so
    g = golden("calibrate_stsb.npz")
    for i in range(int(g["n"])):
        lo, hi = g[f"clip_{i}"]
        y = so.calibrate_mean_var(g[f"in_x_{i}"].copy(), g[f"in_m1_{i}"], g[f"in_v1_{i}"], g[f"in_m2_{i}"], g[f"in_v2_{i}"], lo, hi)
        assert_close(y, g[f"ref_y_{i}"], rtol=2e-7, atol_scale=2e-7, msg=f"case {i}")


# ---- NYUD2 dense FDS variant -------------------------------------------------------------
def test_nyud2_oracle_state_machine(golden):
    from oracle import fds_nyud2_oracle as no
    g = golden("fds_trace_nyud2.npz")
    kw = json.loads(str(g["kw"]))
    O = no.FDSNyud2Oracle(**kw)
    for epoch in range(4):
        xin = g[f"e{epoch}_in_x"].copy()
        y = O.smooth(xin, g[f"e{epoch}_in_labels_b"], epoch)
        assert np.array_equal(xin, g[f"e{epoch}_in_x"])                       # not in place
        assert_close(y, g[f"e{epoch}_ref_smooth"], msg=f"smooth e{epoch}")
        assert_close(O.smooth_grad(g[f"e{epoch}_in_gy"], g[f"e{epoch}_in_labels_b"], epoch), g[f"e{epoch}_ref_gx"], msg=f"grad e{epoch}")
        O.update_last_epoch_stats(epoch)
        O.update_running_stats(g[f"e{epoch}_in_feats"], g[f"e{epoch}_in_labels"], epoch)
        for k in BUFFERS:
            assert_close(getattr(O, k), g[f"e{epoch}_post_{k}"], msg=f"post e{epoch} {k}")
        assert (O.running_mean_last_epoch is O.running_mean) == bool(g[f"e{epoch}_alias"])    # alias broken in this variant
