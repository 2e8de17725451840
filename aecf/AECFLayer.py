"""``aecf.AECFLayer`` module path of the reference (aecf/AECFLayer.py), served by ``aecf_amd.layer``."""
from aecf_amd.layer import *  # noqa: F401,F403
from aecf_amd.layer import __all__, _scaled_dot_product_attention  # noqa: F401
