"""teloscope_amd — MI355X (gfx950) implementation of Teloscope's telomeric-motif scan path.

The product is libteloscan.so (hand-written HIP kernels + C++17 host code behind the C-ABI in
include/teloscan.h).  This package is its thin host-side mirror of the reference interface
(Teloscope.scanSegment, ReadTelomereFilter.matches, expandPatternsWithOrientation).
Importing it without the built library raises ImportError: there is no CPU fallback.
"""
from . import _capi
from ._capi import TeloscanError, build  # noqa: F401

_capi.lib()     # fail loudly at import time if the HIP library is missing

from .teloscope import (ReadTelomereFilter, SegmentData, Teloscope, UserInputTeloscope,  # noqa: E402,F401
                        canonicalOrientation, expandPatternsWithOrientation, getGCContent,
                        getShannonEntropy, revCom)

__all__ = ["Teloscope", "ReadTelomereFilter", "UserInputTeloscope", "SegmentData",
           "expandPatternsWithOrientation", "canonicalOrientation", "revCom", "getGCContent",
           "getShannonEntropy", "TeloscanError", "build"]
