"""CPU-side tests of the product package: C-ABI surface, host LDS weights (bit-exact), module structure,
loud failure without a GPU, and the world_size-2 (gloo) statistic merge. No HIP kernel is launched here."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close
from oracle import fds_oracle, lds_oracle


def test_cabi_exports_every_declared_symbol():
    from dirhip import _lib
    header = open(os.path.join(ROOT, "include", "dir_hip.h")).read()
    declared = set(re.findall(r"\b(dir_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in dir_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # the product library exports product entry points only: no process-wide switches, no measurement probes or
    # experiment kernels (weak #8: those live in tools/lib/libdir_hip_tools.so, declared in tools/csrc/dir_hip_tools.h)
    import subprocess
    exported = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout.splitlines()
                if " T " in l and l.split()[-1].startswith("dir_")}
    assert exported == declared, exported ^ declared
    assert not [n for n in declared if "_set_" in n or "probe" in n]
    assert "_set_" not in header
    tools_header = open(os.path.join(ROOT, "tools", "csrc", "dir_hip_tools.h")).read()
    tools_declared = set(re.findall(r"\b(dir_[a-z0-9_]+)\s*\(", tools_header))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import toolslib
    assert tools_declared == set(toolslib.SIGNATURES) and not (tools_declared & declared)
    th = ctypes.CDLL(toolslib.LIB_PATH)
    assert all(hasattr(th, n) for n in tools_declared)
    assert _lib.lib().dir_abi_version() == _lib.ABI_VERSION == 4
    assert _lib.lib().dir_error_string(-1) == b"invalid argument"


def test_cabi_argument_errors_without_gpu():
    from dirhip import _lib
    L = _lib.lib()
    assert L.dir_fds_scatter_stats_workspace(191509, 2048, 100) > 191509 * 4
    assert L.dir_fds_scatter_stats_workspace(-1, 2048, 100) == 0
    assert L.dir_weighted_loss_workspace(256) == 0 and L.dir_weighted_loss_workspace(1 << 20) > 0
    # null pointers / bad sizes are rejected before any launch
    assert L.dir_fds_smooth_bins(None, None, None, 5, 100, 2048, None, None, None) == -1
    assert L.dir_weighted_loss(9, None, None, None, 4, 0.0, 1.0, 0, None, None, None, 0, None) == -1
    assert L.dir_lds_weights(None, 5, 121, 1, 0, None, 0, None) == -1


def test_conv_planning_entry_points_without_gpu():
    """Host-side planning of the round-2 kernels (no launches): tile rows of the statistics lists, the patch-kernel switch, the
    all-taps weight-gradient workspace and shape gate, argument checks of the new entry points."""
    from dirhip import _lib
    L = _lib.lib()
    # 128-row tiles ...
    assert L.dir_conv_plan_rows(256, 56, 56, 64, 256, 1, 1, 1, 0, 0, 0) == L.dir_conv_stats_rows(256, 56, 56) == 6272
    assert L.dir_conv_plan_rows(256, 56, 56, 128, 128, 3, 3, 2, 1, 0, 0) == L.dir_conv_stats_rows(256, 28, 28) == 1568
    assert L.dir_conv_plan_rows(256, 7, 7, 512, 512, 3, 3, 1, 1, 0, 0) == 98                     # 7x7 maps: not a patch-kernel shape
    # ... chunks of whole image rows for the patch-staged 3x3 / stride-1 layers (2 / 4 / 7 rows per chunk)
    assert L.dir_conv_plan_rows(256, 56, 56, 64, 64, 3, 3, 1, 1, 0, 0) == 256 * 28
    assert L.dir_conv_plan_rows(256, 28, 28, 128, 128, 3, 3, 1, 1, 0, 0) == 256 * 7
    assert L.dir_conv_plan_rows(64, 14, 14, 256, 256, 3, 3, 1, 1, 0, 0) == 64 * 2
    assert L.dir_conv_plan_rows(0, 56, 56, 64, 64, 3, 3, 1, 1, 0, 0) == 0
    # the kernel is an argument of the launch (no process-wide switch): the same layer through the 128-row tile kernel
    assert L.dir_conv_plan_rows(256, 56, 56, 64, 64, 3, 3, 1, 1, 0, _lib.CONV_TILE_REG) == 6272
    # all-taps weight gradient: 256 partials of [64][9][64] floats whatever the channel count, per-tap form for other shapes
    for c, hw in ((64, 56), (128, 28), (256, 14), (512, 7)):
        assert L.dir_conv_wgrad3x3_workspace(256, hw, hw, c, c) == 256 * 64 * 9 * 64 * 4
    assert L.dir_conv_wgrad3x3_workspace(1, 7, 7, 64, 64) == 1 * 64 * 9 * 64 * 4          # one chunk: one split
    assert L.dir_conv_wgrad3x3_workspace(256, 112, 112, 64, 64) == 0 and L.dir_conv_wgrad3x3_workspace(256, 56, 56, 96, 64) == 0
    assert L.dir_conv_wgrad3x3(None, None, None, 256, 56, 56, 64, 64, None, 0, None) == -1
    assert L.dir_conv_dgrad_ex(None, None, None, None, None, None, None, 1, 8, 8, 64, 64, 1, 1, 0, None, None, None, None, None, None, 0, 0, None) == -1
    assert L.dir_bn_bwd_partials(None, None, None, _lib.DIR_BF16, 64, 64, None, None, None, None, None, None, 0, None, 0, None, 0, None) == -1
    assert L.dir_augment_u8(None, None, None, _lib.DIR_F32, 1, 8, 16, None) == -1
    assert L.dir_conv_wgrad_reduce_splits(None, 1, 4, None, None) == -1


def test_lds_weights_native_bit_exact(golden):
    from dirhip import lds
    g = golden("lds_weights.npz")
    for sname in ("agedb", "synth", "frac", "tiny"):
        labels = g[f"in_labels_{sname}"]
        for ci, spec in enumerate(g["configs"]):
            rw, use_lds, k, ks, s = str(spec).split(",")
            w = lds.prepare_weights(labels, rw, lds=bool(int(use_lds)), lds_kernel=k, lds_ks=int(ks), lds_sigma=float(s))
            ref = g[f"ref_w_{sname}_{ci}"]
            if rw == "none":
                assert w is None
                continue
            assert type(w) is list and type(w[0]) is np.float32
            assert np.array_equal(np.asarray(w), ref), (sname, str(spec))


def test_lds_weights_native_sizes_vs_oracle():
    """numpy's chunked pairwise float32 sum (8192-element buffer) across many n."""
    from dirhip import ops
    rng = np.random.default_rng(0)
    for n in list(range(1, 140)) + [8191, 8192, 8193, 16385, 50001, 191509]:
        labels = rng.integers(0, 125, n)
        for rw, use_lds in (("sqrt_inv", True), ("inverse", True), ("sqrt_inv", False)):
            win = lds_oracle.get_lds_kernel_window("gaussian", 5, 2) if use_lds else None
            got = ops.lds_weights(labels, 121, rw, use_lds, win)
            want = lds_oracle.prepare_weights(labels, rw, lds=use_lds)
            assert np.array_equal(got, want), (n, rw, use_lds)


def test_lds_asymmetric_window_matches_scipy():
    from scipy.ndimage import convolve1d
    from dirhip import ops
    rng = np.random.default_rng(1)
    labels = rng.integers(0, 121, 4000)
    win = np.array([0.2, 0.5, 1.0, 0.7, 0.1])
    got = ops.lds_weights(labels, 121, "sqrt_inv", True, win)
    sm = convolve1d(np.sqrt(np.bincount(labels, minlength=121)), weights=win, mode="constant")
    w = (1.0 / sm[labels]).astype(np.float32)
    assert np.array_equal(got, (len(w) / np.sum(w)) * w)


def test_windows_match_golden(golden):
    from dirhip.fds import FDS
    from dirhip.utils import get_lds_kernel_window
    g = golden("windows.npz")
    for i, spec in enumerate(g["grid"]):
        k, ks, s = str(spec).split(",")
        assert np.array_equal(FDS._get_kernel_window(k, int(ks), float(s)).cpu().numpy(), g[f"ref_fds_{i}"])
        assert np.array_equal(np.asarray(get_lds_kernel_window(k, int(ks), float(s))), g[f"ref_lds_{i}"])


def test_fds_module_structure_matches_reference():
    from dirhip.fds import FDS
    f = FDS(64, bucket_num=100, bucket_start=3)
    sd = f.state_dict()
    assert list(sd.keys()) == list(fds_oracle.FDSOracle.BUFFERS)
    assert sd["running_mean"].shape == (97, 64) and sd["num_samples_tracked"].shape == (97,)
    assert sd["epoch"].shape == (1,) and float(sd["epoch"]) == 0.0
    assert float(sd["running_var"].min()) == 1.0 and float(sd["running_mean"].abs().max()) == 0.0
    assert all(v.dtype == torch.float32 for v in sd.values())
    assert (f.start_smooth, f.start_update, f.momentum, f.half_ks) == (1, 0, 0.9, 2)


def test_no_cpu_fallback():
    from dirhip import _lib, loss
    from dirhip.fds import FDS
    from dirhip.utils import calibrate_mean_var
    f = FDS(8, bucket_num=10)
    x, lab = torch.randn(4, 8), torch.ones(4, 1)
    assert f.smooth(x, lab, 0) is x                       # epoch < start_smooth: identity, no kernel
    with pytest.raises(_lib.DirHipError):
        f.smooth(x, lab, 1)
    with pytest.raises(_lib.DirHipError):
        f.update_running_stats(x, lab[:, 0], 0)
    with pytest.raises(_lib.DirHipError):
        loss.weighted_l1_loss(x[:, :1], lab, lab)
    with pytest.raises(_lib.DirHipError):
        calibrate_mean_var(x, x[0], x[0], x[0], x[0])


def test_resnet50_matches_reference_golden(golden):
    """Architecture, state_dict keys and init stream of resnet.py (same torch RNG stream as the reference)."""
    from dirhip.resnet import resnet50
    g = golden("resnet50_forward.npz")
    torch.manual_seed(1234)
    m = resnet50(fds=True, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian",
                 ks=5, sigma=2, momentum=0.9)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 23510081
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["ref_param_sums"], rtol=1e-11, atol=1e-11)
    # the forward itself is GPU-only (fused HIP BatchNorm nodes): tests/test_hip_bn.py checks it against the golden
    from dirhip import _lib
    with pytest.raises(_lib.DirHipError):
        m(torch.randn(1, 3, 224, 224))


# ---- world_size 2 over gloo: the per-epoch FDS statistic exchange -------------------------
def _merge_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dirhip.fds import merge_stats_across_ranks
    d = np.load(os.path.join(tmp, "in.npz"))
    feats, bins, nb = d["feats"], d["bins"], int(d["nb"])
    mine = slice(rank, None, world)
    f, b = feats[mine].astype(np.float64), bins[mine]
    c = feats.shape[1]
    count, mean, m2 = np.zeros(nb), np.zeros((nb, c)), np.zeros((nb, c))
    for k in range(nb):                                   # the per-rank statistics K2 would produce
        rows = f[b == k]
        count[k] = len(rows)
        if len(rows):
            mean[k] = rows.mean(0)
            m2[k] = ((rows - rows.mean(0)) ** 2).sum(0)
    out = merge_stats_across_ranks(torch.tensor(count), torch.tensor(mean), torch.tensor(m2))
    np.savez(os.path.join(tmp, f"out{rank}.npz"), count=out[0].numpy(), mean=out[1].numpy(), m2=out[2].numpy())
    dist.destroy_process_group()


def test_stat_merge_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    rng = np.random.default_rng(9)
    n, c, nb = 700, 12, 9
    bins = rng.integers(0, nb, n)
    bins[bins == 4] = 5                                      # an empty bin
    bins[bins == 7] = 6
    bins[10] = 7                                             # a bin that only rank 0 sees, one row
    feats = rng.normal(0.5, 0.2, (n, c)).astype(np.float32)
    feats[:, 2] = 0.37109375                                 # constant column -> exact zero M2 after the merge
    np.savez(tmp_path / "in.npz", feats=feats, bins=bins, nb=nb)
    port = 29500 + int(rng.integers(0, 2000))
    mp.spawn(_merge_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    o0, o1 = np.load(tmp_path / "out0.npz"), np.load(tmp_path / "out1.npz")
    for k in ("count", "mean", "m2"):
        assert np.array_equal(o0[k], o1[k]), f"ranks disagree on {k}"      # bit-identical tables on every rank
    f64 = feats.astype(np.float64)
    for k in range(nb):
        rows = f64[bins == k]
        assert o0["count"][k] == len(rows)
        if len(rows):
            assert_close(o0["mean"][k], rows.mean(0), rtol=1e-12, atol_scale=1e-13)
            assert_close(o0["m2"][k], ((rows - rows.mean(0)) ** 2).sum(0), rtol=1e-9, atol_scale=1e-12)
            assert o0["m2"][k, 2] == 0.0 and o0["mean"][k, 2] == 0.37109375
        else:
            assert not o0["mean"][k].any() and not o0["m2"][k].any()
