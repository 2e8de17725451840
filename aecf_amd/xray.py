"""Counterpart of the reference's only end-to-end caller of the fusion path:
``AECFModel`` of ``xrays/train_xrays_example.py:108-237`` and one optimisation step of
``train_both_models`` (``:312-377``), with the fusion done by the HIP path.

Encoders / projections / classifier are plain ``torch.nn.Linear`` layers exactly as in the reference (they are
outside the hot path); presence routing, the gather of both-present rows into the pool and the scatter of the
fused rows are the glue either side of the kernel.  Construction order (hence parameter initialisation under a
seed and state_dict keys) is the reference's.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .layer import CurriculumMasking, MultimodalAttentionPool, _DTYPES, _ptr, _stream
from . import _lib, dp


def modality_frontend(features: torch.Tensor, drop: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Missing-modality front-end of one modality in ONE kernel pass: rows with
    ``drop[r]`` set are zeroed (the reference's ``clone()`` + masked write, ``xrays/train_xrays_example.py:173-176``)
    and ``present[r] = ||row|| > 1e-6`` of the row as written (the reference's presence test, ``:202-203``).
    Returns ``(features_out, present_bool)``; without ``drop`` the input tensor itself is returned (no copy).
    Features are data (the reference feeds precomputed CLIP embeddings): no gradient flows to them."""
    if features.device.type != "cuda":
        raise RuntimeError("aecf_amd: the HIP path needs tensors on a ROCm device (no CPU fallback is provided)")
    if features.dim() != 2 or features.dtype not in _DTYPES:
        raise ValueError("modality_frontend expects a [rows, dim] float32 or bfloat16 tensor")
    lib = _lib.load()
    f = features.detach().contiguous()
    rows, dim = f.shape
    present = torch.empty(rows, dtype=torch.uint8, device=f.device)
    d8 = None if drop is None else drop.to(device=f.device, dtype=torch.uint8).contiguous()
    out = f if drop is None else torch.empty_like(f)
    _lib.check(lib.aecf_modality_frontend(rows, dim, _DTYPES[f.dtype], _ptr(f), _ptr(d8),
                                          None if drop is None else _ptr(out), _ptr(present), _stream()),
               "aecf_modality_frontend")
    return out, present.bool()


class AECFModel(nn.Module):
    """ref xrays/train_xrays_example.py:108-237 (same attribute names, same forward contract)."""

    def __init__(self, image_dim: int = 512, text_dim: int = 512, num_classes: int = 80, hidden_dim: int = 256):
        super().__init__()
        self.name = "AECF_Model"
        self.hidden_dim = hidden_dim
        self.curriculum_enabled = False
        self.missing_modality_training = False
        self.image_encoder = nn.Sequential(nn.Linear(image_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.1))   # ref :119-123
        self.text_encoder = nn.Sequential(nn.Linear(text_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.1))     # ref :125-129
        self.curriculum_masking = CurriculumMasking(base_mask_prob=0.15)                                     # ref :132
        self.attention_pool = MultimodalAttentionPool(embed_dim=hidden_dim, num_heads=4,                     # ref :133-138
                                                      curriculum_masking=None, batch_first=True)
        self.fusion_query = nn.Parameter(torch.randn(1, 1, hidden_dim) * 0.02)                               # ref :139
        self.image_proj = nn.Linear(hidden_dim, hidden_dim * 2)                                              # ref :142-143
        self.text_proj = nn.Linear(hidden_dim, hidden_dim * 2)
        self.fusion_proj = nn.Linear(hidden_dim, hidden_dim * 2)                                             # ref :146
        self.classifier = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim), nn.ReLU(), nn.Dropout(0.1),   # ref :149-154
                                        nn.Linear(hidden_dim, num_classes))

    def _simulate_missing_modalities(self, image_features, text_features, missing_prob: float = 0.3):
        """ref :156-172 -- which rows lose which modality (never both); same RNG consumption as the reference.
        The zeroing itself (:173-176) happens in ``modality_frontend`` together with the presence test."""
        if not (self.training and self.missing_modality_training):
            return None, None
        batch_size = image_features.size(0)
        mask_image = torch.rand(batch_size, device=image_features.device) < missing_prob
        mask_text = torch.rand(batch_size, device=text_features.device) < missing_prob
        both_masked = mask_image & mask_text
        if both_masked.any():
            keep_image = torch.rand(int(both_masked.sum()), device=image_features.device) > 0.5
            mask_image[both_masked] = ~keep_image
            mask_text[both_masked] = keep_image
        return mask_image, mask_text

    def toggle_curriculum(self, enabled: bool) -> None:
        """ref :179-187 (without the prints)."""
        self.curriculum_enabled = enabled
        self.attention_pool.curriculum_masking = self.curriculum_masking if enabled else None

    def forward(self, image_features, text_features, return_info: bool = False):
        """ref :189-237."""
        batch_size = image_features.size(0)
        info: Dict[str, torch.Tensor] = {}
        drop_img, drop_txt = self._simulate_missing_modalities(image_features, text_features)
        image_features, img_present = modality_frontend(image_features, drop_img)          # ref :173-176 + :202
        text_features, txt_present = modality_frontend(text_features, drop_txt)           # ref :173-176 + :203
        img_encoded = self.image_encoder(image_features)
        txt_encoded = self.text_encoder(text_features)
        both_present = img_present & txt_present
        only_img = img_present & ~txt_present
        only_txt = ~img_present & txt_present
        fused_features = torch.zeros(batch_size, self.hidden_dim * 2, device=image_features.device,
                                     dtype=img_encoded.dtype)
        if both_present.any():                                                  # ref :212-226
            indices = torch.where(both_present)[0]
            modalities = torch.stack([img_encoded[indices], txt_encoded[indices]], dim=1)
            query = self.fusion_query.expand(len(indices), -1, -1)
            attn_output, attn_info = self.attention_pool(query=query, key=modalities, value=modalities,
                                                         return_info=True)
            fused_features[indices] = self.fusion_proj(attn_output.squeeze(1))
            if return_info:
                info.update(attn_info)
        if only_img.any():                                                      # ref :228-234
            indices = torch.where(only_img)[0]
            fused_features[indices] = self.image_proj(img_encoded[indices])
        if only_txt.any():
            indices = torch.where(only_txt)[0]
            fused_features[indices] = self.text_proj(txt_encoded[indices])
        logits = self.classifier(fused_features)
        return (logits, info) if return_info else logits


def train_step(model: AECFModel, optimizer: torch.optim.Optimizer, criterion: nn.Module, images: torch.Tensor,
               texts: torch.Tensor, labels: torch.Tensor, bucket: Optional[dp.FlatGradBucket] = None,
               ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """One optimisation step of ref :360-377 (zero_grad -> forward(return_info=True) -> BCE -> backward -> step).
    With a FlatGradBucket (data parallel) the gradients of all ranks are averaged by one all-reduce before
    the optimizer step."""
    if bucket is None:
        optimizer.zero_grad(set_to_none=True)
    else:
        bucket.zero()
    logits, info = model(images, texts, return_info=True)
    loss = criterion(logits, labels)
    loss.backward()
    if bucket is not None:
        bucket.all_reduce(average=True)
    optimizer.step()
    return loss.detach(), info
