"""Importable alias of the hyphenated package directory `physical-interaction-video-prediction_amd/`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('physical-interaction-video-prediction_amd')
sys.modules[__name__] = _pkg
