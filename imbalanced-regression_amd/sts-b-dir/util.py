"""Drop-in for the FDS-related part of the reference's ``sts-b-dir/util.py`` (``calibrate_mean_var``, clip [0.5, 2]).
The AllenNLP model / trainer helpers of that file are out of scope."""
import _path  # noqa: F401
from dirhip.fds_stsb import calibrate_mean_var  # noqa: F401
