"""Drop-in for the FDS-related part of the reference's ``nyud2-dir/util.py`` (``calibrate_mean_var``, clip [0.2, 5])."""
import _path  # noqa: F401
from dirhip.fds_nyud2 import calibrate_mean_var  # noqa: F401
