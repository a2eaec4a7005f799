"""Round-6 host tests (no GPU): the bench line's size contract."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fat_result():
    blob = "x" * 400
    fl = {"nominal": {"total_ms": 9.03, "conv_fwd_dgrad_wgrad_ms": 4.5}, "measured_on_this_box": {"total_ms": 12.0}}
    rows = [{"kernel": f"kernel {i} " + blob[:60], "bound": "hbm", "shape": "N=191509 C=2048 Nb=100", "ms": 0.29712345678, "algorithmic_bytes": 1.57e9,
             "achieved": 5281.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.66123456789, "frac_of_measured_peak": 0.9, "note": blob} for i in range(20)]
    return {"metric": "images/sec ResNet-50+FDS IMDB-WIKI 224x224 (train loop incl. FDS epoch tail)", "value": 10233.123456789, "unit": "images/sec", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 25.0123456789, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic", "config": {"workload": blob, "per_gpu_batch": 256, "global_batch": 256, "parallelism": "dp1", "final_loss": 7.123456789},
            "train_only_images_per_sec": 13589.123456789,
            "roofline": {"bound": "mfma", "kernel": blob[:150], "achieved": 492.1026226247471, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.19684104904989883,
                         "traffic": 208146326.20408162, "algorithmic_bytes_per_launch": 202819542.2, "traffic_detail": {"source": blob, "a": 1.0}, "method": blob,
                         "launches_per_step": 113.0, "avg_launch_us": 73.09, "ms_per_step_in_this_kernel": 8.259214750000012, "mfma_busy": 0.288},
            "roofline_step": {"achieved_ms": 18.84, "floor": fl, "frac_of_nominal_floor": 0.48, "frac_of_measured_peak_floor": 0.64, "note": blob},
            "kernel_rooflines": rows, "conv_layers": {"rows": [[1, 2, 3, 4, 5, 6, "fwd", 1.0, 2.0, 1]] * 200}, "peaks": {"a": 1.0}, "input_pipeline": {"x": blob * 10},
            "cpu_baseline": {"value": 8.13, "unit": "images/sec", "cores": 64, "kind": "port", "batch": 8, "sample": blob[:200], "micro": [{"what": blob}] * 8,
                             "batch_note": blob},
            "comm": {"rccl_ranks": 8, "backend": "nccl", "reduce_op": "AVG", "buckets": [{"MB": 32.0, "allreduce_ms": 0.5, "bus_GBs": 100.123456}] * 3,
                     "allreduce_ms_per_step_if_serial": 1.5, "exposed_comm_ms_per_step": 0.2, "note": blob},
            "detail": "gpurun_out/bench_detail.json"}


def test_driver_line_is_short_and_keeps_the_contract_keys():
    """the r05 line (20.5 KB) was not parsed by the driver. Whatever the legs return, the printed line stays under 8 KB, is valid
    JSON on one line, and carries the contract's keys + roofline + cpu_baseline."""
    import bench
    line = bench.driver_line(_fat_result())
    assert len(line) < 8192 and len(line) <= bench.MAX_LINE_BYTES, len(line)
    assert "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["config"]["workload"] and "model" not in d["config"]
    assert "conv_layers" not in d and "input_pipeline" not in d and "micro" not in d["cpu_baseline"]
    assert len(d["kernel_rooflines"]) <= 7
    assert d["value"] == 10233.0 or abs(d["value"] - 10233.123456789) < 1.0          # 5 significant digits


def test_bench_extras_live_outside_bench_py():
    """bench.py is the timed loop + the line; the minutes-long probes are tools/bench_extras.py."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name in ("def conv_layer_probe", "def input_pipeline_probe", "def library_baseline", "def float32_mode_probe"):
        assert name not in src
        assert name in open(os.path.join(ROOT, "tools", "bench_extras.py")).read()


def test_value_groups_equal_the_reference_row_groups():
    """SURVEY A.8 host logic: dirhip.fds.value_groups (the row grouping of update_running_stats for non-integer labels) against the oracle's restatement
    of the reference's torch.unique loop (oracle/fds_oracle.py:_row_groups, fds.py:91-99), boundary values present / absent, multi-rank label pools."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
    from dirhip.fds import value_groups
    from oracle.fds_oracle import _row_groups
    rng = np.random.default_rng(0)
    for trial in range(40):
        start, num = int(rng.integers(0, 4)), int(rng.integers(8, 30))
        step = float(rng.choice([0.5, 0.25, 0.125, 1.0]))
        labels = (np.round(rng.normal((start + num) / 2, num / 3, 300) / step) * step).astype(np.float32)
        if trial % 3 == 0:
            labels = labels[(labels != start) & (labels != num - 1)]              # boundary values absent: nothing lumps the out-of-range rows
        elif trial % 3 == 1:
            labels[:2] = start; labels[2:4] = num - 1
        gid, bin_ptr, ng = value_groups(labels, start, num)
        exp_gid = np.full(labels.shape, -1, np.int32)
        exp_bins = []
        for g, (v, rows) in enumerate(_row_groups(labels, start, num)):
            exp_gid[rows] = g
            exp_bins.append(int(np.float32(v) - np.float32(start)))
        assert ng == len(exp_bins) and np.array_equal(gid, exp_gid), trial
        nb = num - start
        assert bin_ptr.shape == (nb + 1,) and bin_ptr[0] == 0 and bin_ptr[-1] == ng
        for g, b in enumerate(exp_bins):
            assert bin_ptr[b] <= g < bin_ptr[b + 1], (trial, g, b)
        # a second "rank" contributes values this rank does not hold: the group list is the union's, this rank's rows index into it
        other = (np.round(rng.normal((start + num) / 2, num / 3, 100) / step) * step).astype(np.float32)
        both = np.concatenate([labels, other])
        gid2, bin_ptr2, ng2 = value_groups(labels, start, num, all_labels=both)
        gid_all, _, ng_all = value_groups(both, start, num)
        assert ng2 == ng_all and np.array_equal(gid2, gid_all[:labels.size])


def test_float32_arithmetic_scope_is_thread_local_nested_and_restored():
    """conv_f32.arithmetic(name): the arithmetic of the float32 conv nodes built inside the scope (round 6, ABI 4: exact / split-bf16 x3 / x2). No process-wide
    switch: the scope is thread-local, nests, and is restored when the block is left by an exception; the engine / CLI only accept the three names."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "imbalanced-regression_amd"))
    import pytest
    from dirhip import conv_f32 as c
    assert c.current_arith() == 0 and c.ARITHMETICS == {"exact": 0, "x3": 3, "x2": 4}
    with c.arithmetic("x3"):
        assert c.current_arith() == c.TILE_X3
        with c.arithmetic("x2"):
            assert c.current_arith() == c.TILE_X2
            with c.arithmetic(None):
                assert c.current_arith() == 0
            assert c.current_arith() == c.TILE_X2
        assert c.current_arith() == c.TILE_X3
        seen = []
        t = threading.Thread(target=lambda: seen.append(c.current_arith()))       # another thread (autograd's, a prefetcher's) is outside the scope
        t.start(); t.join()
        assert seen == [0]
    with pytest.raises(RuntimeError):
        with c.arithmetic("x2"):
            raise RuntimeError("boom")
    assert c.current_arith() == 0
    with pytest.raises(KeyError):
        with c.arithmetic("x4"):
            pass
    # header, ctypes table and CLI agree on the new names
    hdr = open(os.path.join(ROOT, "include", "dir_hip.h")).read()
    assert "#define DIR_CONV_F32_TILE_X3 3" in hdr and "#define DIR_CONV_F32_TILE_X2 4" in hdr and "#define DIR_ABI_VERSION 4" in hdr
    from dirhip import _lib
    assert "dir_conv_f32_fwd_stats_variant" in _lib.SIGNATURES and _lib.ABI_VERSION == 4
    src = open(os.path.join(ROOT, "imbalanced-regression_amd", "dirhip", "train_main.py")).read()
    assert "choices=['bf16', 'fp32', 'fp32x3', 'fp32x2']" in src and "'--amp_early'" in src
    from dirhip.parallel import DataParallelEngine
    import torch
    eng = DataParallelEngine(torch.nn.Linear(2, 2), amp_dtype=None, f32_arith="x2")
    assert eng.f32_arith == "x2"
    eng.set_amp_dtype(torch.bfloat16)
    assert eng.f32_arith == "x2" and eng.amp_dtype == torch.bfloat16            # the float32 arithmetic is kept across a bf16 phase
    eng.set_amp_dtype(None, f32_arith="exact")
    assert eng.f32_arith == "exact"
    with pytest.raises(AssertionError):
        DataParallelEngine(torch.nn.Linear(2, 2), f32_arith="x4")
