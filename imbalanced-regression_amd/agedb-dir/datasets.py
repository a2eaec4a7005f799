"""Drop-in for the reference's ``agedb-dir/datasets.py``."""
import _path  # noqa: F401
from dirhip.datasets import AgeDB, IMDBWIKI, SyntheticAgeDataset  # noqa: F401
from dirhip.utils import get_lds_kernel_window  # noqa: F401
