"""xrsfm_amd — MI355X-native bundle adjustment for XRSfM's src/optimization path.

Product code = the HIP library behind include/xrsfm_ba.h (xrsfm_amd/csrc) and
this thin ctypes mirror of the C-ABI.  No CPU fallback exists on purpose.
"""
from . import _build, capi, synth  # noqa: F401
from .capi import Context, ProblemArrays, default_options, solve  # noqa: F401
