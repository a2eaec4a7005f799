"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference losses.

Follows ``imdb-wiki-dir/loss.py:5-48`` (= ``agedb-dir/loss.py``). Forward values are
computed op by op in float32 like the reference; gradients w.r.t. ``inputs`` are the
closed forms of what torch autograd produces for those expressions (float64 arithmetic,
rounded to float32). Pinned by ``tests/test_oracle_golden.py`` against golden vectors
generated from the reference (values and autograd gradients).
"""
import numpy as np

F32 = np.float32
KINDS = ("mse", "l1", "focal_mse", "focal_l1", "huber")


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def weighted_loss(kind, inputs, targets, weights=None, activate="sigmoid", beta=None, gamma=1):
    """Returns (loss: float32 scalar, dloss/dinputs: float32 array like inputs)."""
    x = np.asarray(inputs, dtype=np.float32)
    y = np.asarray(targets, dtype=np.float32)
    w = None if weights is None else np.broadcast_to(np.asarray(weights, dtype=np.float32), x.shape)
    n = x.size
    d32 = x - y
    d = d32.astype(np.float64)
    ad = np.abs(d)
    sgn = np.sign(d)
    if kind == "mse":                                   # loss.py:5-10
        per = d32 ** 2
        g = 2.0 * d
    elif kind == "l1":                                  # loss.py:13-18
        per = np.abs(d32)
        g = sgn
    elif kind in ("focal_mse", "focal_l1"):             # loss.py:21-38
        beta = 0.2 if beta is None else beta
        if activate == "tanh":
            act32 = np.tanh(F32(beta) * np.abs(d32))
            act = np.tanh(beta * ad)
            dact = (1.0 - act ** 2) * beta
        else:
            act32 = F32(2) * (F32(1) / (F32(1) + np.exp(-(F32(beta) * np.abs(d32))))) - F32(1)
            s = _sigmoid(beta * ad)
            act = 2.0 * s - 1.0
            dact = 2.0 * s * (1.0 - s) * beta
        base32 = d32 ** 2 if kind == "focal_mse" else np.abs(d32)
        per = base32 * (act32 ** F32(gamma))
        base = d ** 2 if kind == "focal_mse" else ad
        dbase = 2.0 * d if kind == "focal_mse" else sgn
        with np.errstate(all="ignore"):
            g = dbase * act ** gamma + base * gamma * act ** (gamma - 1) * dact * sgn
    elif kind == "huber":                               # loss.py:41-48
        beta = 1.0 if beta is None else beta
        a32 = np.abs(d32)
        per = np.where(a32 < F32(beta), F32(0.5) * a32 ** 2 / F32(beta), a32 - F32(0.5 * beta))
        g = np.where(ad < beta, d / beta, sgn)
    else:
        raise ValueError(kind)
    per = per.astype(np.float32)
    if w is not None:
        per = per * w
        g = g * w.astype(np.float64)
    loss = np.mean(per, dtype=np.float32)
    return F32(loss), (g / n).astype(np.float32)
