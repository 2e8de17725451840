"""aecf_amd -- MI355X-native AECF fusion path (drop-in for the ``aecf`` package of leochlon/aecf).

Same public surface as ref: aecf/__init__.py:8-21; arithmetic in hand-written HIP (libaecf_hip.so).
"""
from .layer import (
    CurriculumMasking,
    MultimodalAttentionPool,
    multimodal_attention_pool,
    create_fusion_pool,
)

__version__ = "0.1.0"
__all__ = [
    "CurriculumMasking",
    "MultimodalAttentionPool",
    "multimodal_attention_pool",
    "create_fusion_pool",
]
