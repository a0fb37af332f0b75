"""Fold eval-mode BatchNorm into the convolution registered immediately before it in the same parent module.

Same rule as the reference's ``fuse_bn_recursively`` (semantic_segmentation/lib/utils/bn_fusion.py:6-74): it looks
at *registration order* inside each parent, so encoder ``conv -> bn`` pairs fold while the decoder's
``norm -> relu -> conv`` blocks (BN first) do not, and training-mode BNs (the online-trained policy net) are skipped.
The fold therefore defines the op sequence the block engine sees (21 padded ops for SwiftNet-RN18)."""
from __future__ import annotations

import torch
import torch.nn as nn


@torch.no_grad()
def _fold_pair(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    w = conv.weight
    bias = conv.bias if conv.bias is not None else torch.zeros(w.size(0), dtype=w.dtype, device=w.device)
    gamma = bn.weight if bn.weight is not None else torch.ones_like(bn.running_mean)
    beta = bn.bias if bn.bias is not None else torch.zeros_like(bn.running_mean)
    denom = torch.sqrt(bn.running_var + bn.eps)
    shift = beta - gamma.mul(bn.running_mean).div(denom)
    scale = gamma.div(denom)
    new_bias = bias.detach().clone().mul_(scale).add_(shift)
    w.mul_(scale.view(-1, 1, 1, 1))
    if conv.bias is None:
        conv.bias = nn.Parameter(new_bias)
    else:
        conv.bias.copy_(new_bias)


def fold_batchnorm(model: nn.Module, verbose: bool = False) -> nn.Module:
    for parent_name, parent in model.named_modules():
        children = list(parent.named_children())
        for i in range(1, len(children)):
            name, m = children[i]
            prev = children[i - 1][1]
            if not isinstance(m, nn.BatchNorm2d) or m.training or getattr(m, "fused", False):
                continue
            if isinstance(prev, nn.Conv2d) and not getattr(prev, "fused", False):
                _fold_pair(prev, m)
                prev.fused = True
                setattr(parent, name, nn.Identity())
                if verbose:
                    print(f"BatchNorm fused with Conv: {parent_name}.{name}")
    return model
