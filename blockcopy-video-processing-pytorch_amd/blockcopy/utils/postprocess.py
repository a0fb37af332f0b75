"""Post-processing helpers for callers of a block-executed segmentation model (NOT part of the reference's API: an addition).

``upsample_argmax(logits, size)`` is the prediction map the reference's driver forms for the labelled frame of a clip --
``F.interpolate(out, size, mode='bilinear')`` followed by ``out.max(dim=1)[1]`` (semantic_segmentation/test_swiftnet.py:190-194) --
in ONE pass over the logits (``bc_upsample_argmax``): at 19 classes and 1024 x 2048 the two library passes write and re-read a
160 MB intermediate that nothing else uses.  Same arithmetic (ATen's bilinear source index and weights, rounding to the logits' dtype,
first maximal class), so the predictions are those of the two-op form."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def upsample_argmax(logits: torch.Tensor, size, align_corners: bool = False) -> torch.Tensor:
    """int64 (N, H, W) class map of ``logits`` (N, C, h, w) at resolution ``size``."""
    if logits.dim() != 4:
        raise ValueError(f"upsample_argmax expects (N, C, h, w) logits, got {tuple(logits.shape)}")
    if isinstance(size, int):
        size = (size, size)
    if logits.is_cuda:
        from blockcopy.backend import get_backend      # (raises if the HIP library is missing: no silent fallback on a GPU)

        if hasattr(logits, "_plain"):
            logits = logits._plain()
        return get_backend().upsample_argmax(logits.detach(), size, align_corners)
    # host tensors: the two stock ops
    return F.interpolate(logits, size=tuple(size), mode="bilinear", align_corners=align_corners).max(dim=1)[1]
