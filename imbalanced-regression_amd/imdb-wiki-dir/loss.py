"""Drop-in for the reference's ``imdb-wiki-dir/loss.py`` (star-imported by train.py:15; also re-exports torch / F)."""
import _path  # noqa: F401
import torch  # noqa: F401
import torch.nn.functional as F  # noqa: F401
from dirhip.loss import (weighted_focal_l1_loss, weighted_focal_mse_loss, weighted_huber_loss,  # noqa: F401
                         weighted_l1_loss, weighted_mse_loss)
