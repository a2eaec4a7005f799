"""Drop-in for the reference's ``imdb-wiki-dir/utils.py`` (star-imported by train.py:17, which relies on it for the names
torch / np / os / logging / shutil as well — SURVEY.md §8b)."""
import _path  # noqa: F401
import os  # noqa: F401
import shutil  # noqa: F401
import torch  # noqa: F401
import logging  # noqa: F401
import numpy as np  # noqa: F401
from scipy.ndimage import gaussian_filter1d  # noqa: F401
from scipy.signal.windows import triang  # noqa: F401
from dirhip.utils import (AverageMeter, ProgressMeter, adjust_learning_rate, calibrate_mean_var,  # noqa: F401
                          get_lds_kernel_window, prepare_folders, query_yes_no, save_checkpoint)
