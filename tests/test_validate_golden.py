"""`validate` / `shot_metrics` (imdb-wiki-dir/train.py:286-391) against golden outputs of the REFERENCE's own functions
(tests/golden/validate.npz, produced by tests/golden/gen_golden_r2.py, which compiles the two functions out of the
reference's train.py with `ast`): overall MSE / L1 / G-Mean and the many / median / low-shot table on seeded predictions.
This is the executable form of north_star's "val-MAE parity" when the image data is absent: identical predictions ->
identical reported numbers. Also: adjust_learning_rate, save_checkpoint layout, the non-destructive prepare_folders."""
import os
import types

import numpy as np
import pytest
import torch


class _Echo(torch.nn.Module):
    def forward(self, inputs):
        return inputs.reshape(inputs.shape[0], -1)[:, :1].clone()


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_validate_and_shot_metrics_match_reference_functions(golden, ci):
    from dirhip.train_main import shot_metrics, validate
    g = golden("validate.npz")
    preds, labels, bs = g[f"in_preds_{ci}"], g[f"in_labels_{ci}"], int(g[f"batch_{ci}"])
    train_labels = g["in_train_labels"]

    def batches():
        for i in range(0, len(labels), bs):
            inp = torch.zeros(len(labels[i:i + bs]), 3, 2, 2)
            inp[:, 0, 0, 0] = torch.tensor(preds[i:i + bs])
            yield inp, torch.tensor(labels[i:i + bs]).view(-1, 1), torch.ones(len(labels[i:i + bs]), 1)
    args = types.SimpleNamespace(print_freq=3)
    mse, l1, gm = validate(batches, (len(labels) + bs - 1) // bs, _Echo(), args, train_labels=train_labels)
    ref = g[f"ref_validate_{ci}"]
    np.testing.assert_allclose([mse, l1, gm], ref, rtol=1e-6, atol=0)
    assert mse == ref[0] and l1 == ref[1]                       # same per-batch float32 means, same float64 weighted average
    sd = shot_metrics(preds, labels, train_labels)
    got = np.array([[sd[s][k] for k in ("mse", "l1", "gmean")] for s in ("many", "median", "low")], dtype=np.float64)
    np.testing.assert_allclose(got, g[f"ref_shot_{ci}"], rtol=1e-6, atol=0)
    # tensor inputs take the same path (train.py:341-343)
    sd_t = shot_metrics(torch.tensor(preds), torch.tensor(labels), train_labels)
    assert sd_t["low"]["l1"] == sd["low"]["l1"]
    with pytest.raises(TypeError):
        shot_metrics(list(preds), labels, train_labels)


def test_adjust_learning_rate_and_checkpoint_layout(tmp_path):
    from dirhip.utils import adjust_learning_rate, save_checkpoint
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    args = types.SimpleNamespace(lr=1e-3, schedule=[60, 80], store_root=str(tmp_path), store_name="run")
    got = []
    for epoch in (0, 59, 60, 79, 80, 89):
        adjust_learning_rate(opt, epoch, args)
        got.append(opt.param_groups[0]["lr"])
    np.testing.assert_allclose(got, [1e-3, 1e-3, 1e-4, 1e-4, 1e-5, 1e-5], rtol=1e-12)      # utils.py:81-86
    os.makedirs(tmp_path / "run")
    state = {"epoch": 3, "model": "resnet50", "best_loss": 7.5, "state_dict": {"module.linear.bias": torch.ones(1)}, "optimizer": opt.state_dict()}
    save_checkpoint(args, state, is_best=False)
    assert os.path.isfile(tmp_path / "run" / "ckpt.pth.tar") and not os.path.isfile(tmp_path / "run" / "ckpt.best.pth.tar")
    save_checkpoint(args, state, is_best=True)
    best = torch.load(tmp_path / "run" / "ckpt.best.pth.tar")
    assert set(best) == {"epoch", "model", "best_loss", "state_dict", "optimizer"} and best["epoch"] == 3     # train.py:209-215


def test_prepare_folders_never_deletes_without_an_explicit_yes(tmp_path, monkeypatch):
    """a non-interactive relaunch of the same configuration must not silently delete the previous run's
    checkpoints. Without a terminal the answer is no; --overwrite is the explicit opt-in; --resume keeps the folder."""
    import io
    import sys
    from dirhip.utils import prepare_folders, query_yes_no
    store = tmp_path / "ckpt" / "run"
    os.makedirs(store)
    (store / "ckpt.pth.tar").write_bytes(b"precious")
    monkeypatch.setattr(sys, "stdin", io.StringIO(""))                        # no tty
    assert query_yes_no("overwrite?") is False
    args = types.SimpleNamespace(store_root=str(tmp_path / "ckpt"), store_name="run", resume="", pretrained="", evaluate=False, overwrite=False)
    with pytest.raises(RuntimeError, match="--overwrite"):
        prepare_folders(args)
    assert (store / "ckpt.pth.tar").read_bytes() == b"precious"
    args.resume = str(store / "ckpt.pth.tar")
    prepare_folders(args)                                                     # resuming never touches the folder
    assert (store / "ckpt.pth.tar").exists()
    args.resume, args.overwrite = "", True
    prepare_folders(args)
    assert store.is_dir() and not (store / "ckpt.pth.tar").exists()


def test_shard_padding_is_marked_invalid():
    from dirhip.parallel import shard_indices
    n, world = 1003, 4
    got = [shard_indices(n, r, world, epoch_seed=5, with_valid=True) for r in range(world)]
    idx = torch.cat([g[0] for g in got])
    valid = torch.cat([g[1] for g in got])
    assert int(valid.sum()) == n and sorted(idx[valid].tolist()) == list(range(n))     # the valid positions are the dataset, once each
    assert all(len(g[0]) == 251 for g in got)
    i1, v1 = shard_indices(n, 0, 1, with_valid=True)
    assert bool(v1.all()) and torch.equal(i1, torch.arange(n))
