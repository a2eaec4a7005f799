"""NYUD2-DIR per-pixel LDS weights (nyud2-dir/loaddata.py:11-19,29-69) against golden outputs of the
reference's own `depthDataset._get_bucket_weights` / `_get_weights` (tests/golden/nyud2_lds_weights.npz): the 100 bucket
weights bit for bit on the host; the per-pixel weight map bit for bit on the GPU (-m gpu)."""
import numpy as np
import pytest
import torch


def _configs(g):
    for ci, c in enumerate(g["configs"]):
        rw, lds, k, ks, s = str(c).split(",")
        yield ci, rw, bool(int(lds)), k, int(ks), float(s)


def test_bucket_weights_bit_exact_vs_reference(golden):
    from dirhip import lds_nyud2 as M
    g = golden("nyud2_lds_weights.npz")
    assert np.array_equal(np.asarray(M.TRAIN_BUCKET_NUM, dtype=np.int64), g["ref_train_bucket_num"])
    for ci, rw, lds, k, ks, s in _configs(g):
        sig = int(s) if float(s).is_integer() else s
        bw = M.get_bucket_weights(rw, lds, k, ks, sig, bucket_num=100, bucket_start=7)
        assert all(type(x) is np.float32 for x in bw) and len(bw) == 100
        assert np.array_equal(np.asarray(bw, dtype=np.float32), g[f"ref_bucket_weights_{ci}"]), (ci, rw, lds, k, ks, s)
    assert M.get_bucket_weights("none") is None
    with pytest.raises(AssertionError):
        M.get_bucket_weights("none", lds=True)
    assert M.get_bin_idx(np.float32(0.7)) == 7 and M.get_bin_idx(np.float32(10.4)) == 99 and M.get_bin_idx(np.float32(0.69999)) == 6


@pytest.mark.gpu
def test_pixel_weight_map_bit_exact_vs_reference(golden):
    from dirhip import lds_nyud2 as M
    g = golden("nyud2_lds_weights.npz")
    depth = torch.tensor(g["in_depth"]).cuda()
    for ci, rw, lds, k, ks, s in _configs(g):
        pw = M.PixelWeights([np.float32(x) for x in g[f"ref_bucket_weights_{ci}"]])
        w = pw.weights(depth)
        assert w.shape == depth.shape and w.dtype == torch.float32
        assert np.array_equal(w.cpu().numpy(), g[f"ref_pixel_weights_{ci}"]), ci
    assert np.array_equal(M.PixelWeights(None).weights(depth).cpu().numpy(), g["ref_pixel_weights_none"])
