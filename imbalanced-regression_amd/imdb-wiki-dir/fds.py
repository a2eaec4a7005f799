"""Drop-in for the reference's ``imdb-wiki-dir/fds.py``: ``from fds import FDS`` (default bucket_start=0, fds.py:16)."""
import _path  # noqa: F401
from dirhip import fds as _impl
from dirhip.utils import calibrate_mean_var  # noqa: F401  (the reference module imports it too)


class FDS(_impl.FDS):
    def __init__(self, feature_dim, bucket_num=100, bucket_start=0, start_update=0, start_smooth=1,
                 kernel='gaussian', ks=5, sigma=2, momentum=0.9):
        super().__init__(feature_dim, bucket_num, bucket_start, start_update, start_smooth, kernel, ks, sigma, momentum)
