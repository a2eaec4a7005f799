"""Puts the ``dirhip`` package (one directory up) on sys.path for the flat drop-in modules of this folder."""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
