"""Pins oracle/torch_oracle.py (the CPU 'port' used as bench.py's cpu_baseline and as the differentiable
checker of the end-to-end GPU tests) against the LIVE reference: same init stream, same loss trajectory,
same FDS buffers on identical inputs. Build container only (needs /root/reference)."""
import numpy as np
import pytest
import torch

from oracle import torch_oracle


@pytest.mark.needs_reference
def test_port_matches_live_reference_training():
    from oracle import refshim
    torch.set_num_threads(8)
    kw = dict(bucket_num=100, bucket_start=0, start_update=0, start_smooth=1, kernel="gaussian", ks=5, sigma=2, momentum=0.9)
    torch.manual_seed(7)
    ref = refshim.make_resnet50("imdb-wiki-dir", fds=True, **kw)
    torch.manual_seed(7)
    port = torch_oracle.RefResNet50(fds=True, **kw)
    assert list(ref.state_dict().keys()) == list(port.state_dict().keys())
    for (k, a), b in zip(ref.state_dict().items(), port.state_dict().values()):
        assert torch.equal(a, b), k
    refl = refshim.load("imdb-wiki-dir").loss
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3)
    opt_p = torch.optim.Adam(port.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(0)
    batches = [(torch.randn(3, 3, 224, 224, generator=g), torch.tensor([[25.0], [40.0], [25.0 + i]]),
                torch.rand(3, 1, generator=g) + 0.5) for i in range(2)]
    for epoch in range(3):
        for x, y, w in batches:
            ref.train()
            with refshim.cuda_identity():
                out, _ = ref(x, y, epoch)
            loss = refl.weighted_l1_loss(out, y, w)
            opt_r.zero_grad(); loss.backward(); opt_r.step()
            lp = torch_oracle.train_step(port, opt_p, x, y, w, epoch, "l1")
            assert abs(lp - loss.item()) <= 1e-6 * abs(loss.item()), (epoch, lp, loss.item())
        # epoch tail, reference side = train.py:269-281 verbatim on CPU
        enc, lab = [], []
        with torch.no_grad(), refshim.cuda_identity():
            for x, y, _ in batches:
                _, f = ref(x, y, epoch)
                enc.extend(f.data.squeeze().cpu().numpy()); lab.extend(y.data.squeeze().cpu().numpy())
            ref.FDS.update_last_epoch_stats(epoch)
            ref.FDS.update_running_stats(torch.from_numpy(np.vstack(enc)), torch.from_numpy(np.hstack(lab)), epoch)
        torch_oracle.epoch_tail(port, [(x, y) for x, y, _ in batches], epoch)
        for (k, a), b in zip(ref.FDS.state_dict().items(), port.FDS.state_dict().values()):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (epoch, k)
