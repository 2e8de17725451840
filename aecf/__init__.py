"""Drop-in name for the reference package: ``import aecf`` resolves to the MI355X-native implementation.

The reference's public surface (``aecf/__init__.py:8-21`` of leochlon/aecf: ``CurriculumMasking``,
``MultimodalAttentionPool``, ``multimodal_attention_pool``, ``create_fusion_pool``, ``__version__``) is re-exported
from ``aecf_amd``; ``aecf.AECFLayer`` is kept importable because user code imports from the module path too
(``from aecf.AECFLayer import MultimodalAttentionPool``).  Nothing here computes anything.
"""
from aecf_amd import *  # noqa: F401,F403
from aecf_amd import __all__, __version__  # noqa: F401
