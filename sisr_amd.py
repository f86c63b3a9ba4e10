"""Importable alias for the package directory ``super-resolution-meta-attention-networks_amd``
(hyphens cannot appear in an ``import`` statement).  ``import sisr_amd`` returns that package; reach
sub-modules as attributes (``sisr_amd.ops``, ``sisr_amd.handlers``...)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("super-resolution-meta-attention-networks_amd")
