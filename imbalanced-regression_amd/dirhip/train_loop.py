"""The training hot loop of ``imdb-wiki-dir/train.py`` / ``agedb-dir/train.py`` (``train()``, lines 234-283),
restructured for one process per MI355X:

  * ``train_step``   = ``train.py:246-262`` (forward incl. ``FDS.smooth`` -> weighted loss -> backward -> step).
    The loss value is NOT read back every step (the reference's ``loss.item()`` at ``:256-258`` is a host
    sync per step); it is accumulated on the device and checked/read once per ``print_freq`` steps.
  * ``epoch_tail``   = ``train.py:269-281``: the second, no-grad, train-mode pass over this rank's shard
    writes every ``[B, 2048]`` encoding straight into one preallocated device buffer (no ``.cpu().numpy()``
    / ``np.vstack`` / ``.cuda()`` round trip of the whole ``[N, 2048]`` matrix), then
    ``FDS.update_last_epoch_stats`` + ``FDS.update_running_stats`` (statistics merged across ranks).
"""
import torch

from . import loss as losses


def resolve_loss(name):
    return getattr(losses, f"weighted_{name}_loss")          # same name-based lookup as train.py:255


def train_step(model, optimizer, inputs, targets, weights, epoch, loss_fn, fds=True):
    """One optimisation step. Returns the loss as a 0-dim device tensor (no host sync)."""
    outputs = model(inputs, targets, epoch)
    if fds:
        outputs = outputs[0]
    loss = loss_fn(outputs, targets, weights)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()


class EpochFeatures:
    """Device-resident [N, C] float32 buffer + [N] labels for the epoch-end statistics pass."""

    def __init__(self, capacity, feature_dim, device):
        self.features = torch.empty(capacity, feature_dim, dtype=torch.float32, device=device)
        self.labels = torch.empty(capacity, dtype=torch.float32, device=device)
        self.n = 0

    def reset(self):
        self.n = 0

    def append(self, feature, targets):
        b = feature.shape[0]
        if self.n + b > self.features.shape[0]:              # grow geometrically (ragged last batches etc.)
            cap = max(self.n + b, 2 * self.features.shape[0])
            nf = torch.empty(cap, self.features.shape[1], dtype=torch.float32, device=self.features.device)
            nl = torch.empty(cap, dtype=torch.float32, device=self.labels.device)
            nf[:self.n] = self.features[:self.n]
            nl[:self.n] = self.labels[:self.n]
            self.features, self.labels = nf, nl
        self.features[self.n:self.n + b] = feature.reshape(b, -1)
        self.labels[self.n:self.n + b] = targets.reshape(b)
        self.n += b

    def view(self):
        return self.features[:self.n], self.labels[:self.n]


@torch.no_grad()
def epoch_tail(model, batches, epoch, store):
    """``batches`` yields ``(inputs, targets)`` or ``(inputs, targets, valid)`` already on the device (``valid``: HOST bool
    ``[B]``, False for rows that only pad a data-parallel shard — they run through the network with their batch but are
    left out of the statistics). ``model`` is the (possibly wrapped) network in train mode (BN uses batch statistics,
    like the reference's second pass — SURVEY A.2)."""
    fds_mod = model.module.FDS if hasattr(model, "module") else model.FDS
    store.reset()
    for batch in batches:
        inputs, targets = batch[0], batch[1]
        valid = batch[2] if len(batch) > 2 else None
        _, feature = model(inputs, targets, epoch)
        if valid is not None and not bool(valid.all()):          # `valid` is a host tensor: no device sync for full batches
            keep = valid.to(feature.device)
            feature, targets = feature[keep], targets[keep]
        if feature.shape[0]:
            store.append(feature, targets)
    feats, labels = store.view()
    fds_mod.update_last_epoch_stats(epoch)
    fds_mod.update_running_stats(feats, labels, epoch)
