"""Drop-in for the reference's ``agedb-dir/resnet.py``: ``from resnet import resnet50``."""
import _path  # noqa: F401
from dirhip.resnet import Bottleneck, ResNet, resnet50  # noqa: F401
