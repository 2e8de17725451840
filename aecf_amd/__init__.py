"""aecf_amd -- MI355X-native AECF fusion path (drop-in for the ``aecf`` package of leochlon/aecf).

The public names are exactly the ones the reference package exports (ref: aecf/__init__.py:8-21); they are defined in
``aecf_amd.layer`` (host mirror of aecf/AECFLayer.py) and re-exported from its ``__all__`` so that the two lists cannot
drift apart.  The arithmetic runs in hand-written HIP (libaecf_hip.so, include/aecf_hip.h); nothing here falls back to
PyTorch math.
"""
from . import layer as _layer

__all__ = list(_layer.__all__)
globals().update({name: getattr(_layer, name) for name in __all__})

__version__ = "0.1.0"      # the reference's version, whose surface this mirrors
