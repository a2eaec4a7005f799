"""Drop-in for the reference's ``nyud2-dir/models/fds.py``: ``from models.fds import FDS`` (dense per-pixel FDS)."""
try:
    from . import _path  # noqa: F401
except ImportError:
    import _path  # noqa: F401
from dirhip.fds_nyud2 import FDS, calibrate_mean_var  # noqa: F401
