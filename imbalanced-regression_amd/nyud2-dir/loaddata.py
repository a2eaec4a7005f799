"""Drop-in for the LDS part of the reference's ``nyud2-dir/loaddata.py`` (``TRAIN_BUCKET_NUM``, the bucket weights of
``depthDataset._get_bucket_weights`` and the per-pixel weight map of ``_get_weights``, loaddata.py:11-19,29-69). The image
loading / augmentation part of that file belongs to the NYUD2 model stack, which is out of scope."""
import _path  # noqa: F401
from dirhip.lds_nyud2 import TRAIN_BUCKET_NUM, PixelWeights, get_bin_idx, get_bucket_weights  # noqa: F401
