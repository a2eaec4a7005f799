"""Drop-in for the reference's ``sts-b-dir/fds.py``: ``from fds import FDS`` (histogram-edge buckets on [0, 5], bucket_num=50)."""
import _path  # noqa: F401
from dirhip.fds_stsb import FDS, calibrate_mean_var  # noqa: F401
