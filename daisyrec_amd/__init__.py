"""daisyrec_amd — MI355X-native MF + BPR training hot path behind daisyRec's model API.

Importing the package loads ``lib/libdaisyrec_hip.so`` (see ``_native.py``); there
is no CPU fallback.
"""
from . import _native, ops  # noqa: F401
from .model.MFRecommender import MF  # noqa: F401

__version__ = "0.1.0"
