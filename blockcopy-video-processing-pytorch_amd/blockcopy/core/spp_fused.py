"""Fast path for the reference's pyramid-pooling module inside ``blockcopy_noblocks``.

``semantic_segmentation/lib/models/swiftnet/util.py`` ``SpatialPyramidPooling.forward`` (decorated with ``@blockcopy_noblocks``)
runs, after its first block, ``3 x [adaptive_avg_pool2d, BN, ReLU, conv1x1 (42 channels), bilinear upsample] + cat + BN + ReLU +
conv1x1`` on a dense stride-32 map: 15 launches of 4-13 us for < 0.3 GFLOP -- launch latency only (csrc/spp.inc).  When the
module wrapped by the decorator has exactly that structure, the block engine computes the same function in two launches
(``bc_spp_levels_nhwc`` + ``bc_spp_fuse_nhwc``); anything that does not match -- other module, other options, training mode,
gradients, an upsampling function that is not plain bilinear -- takes the generic op-by-op route.  The result differs from that route
by fp32 summation order only (every intermediate rounding to the map's dtype is reproduced)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from . import fusion
from ..backend import get_backend

# the last block computes the executed tiles only and returns them packed (bc_spp_fuse_packed_nhwc); 0 = dense result + gather
PACKED_RESULT = os.environ.get("BLOCKCOPY_SPP_PACKED", "1") != "0"


def _block_parts(block):
    """(bn | None, conv) of a BN -> ReLU -> conv1x1 block, or None if the block is anything else."""
    if not isinstance(block, nn.Sequential):
        return None
    kids = list(block.named_children())
    names = [n for n, _ in kids]
    mods = dict(kids)
    if names not in (["norm", "relu", "conv"], ["relu", "conv"]):
        return None
    bn, conv = mods.get("norm"), mods["conv"]
    if not isinstance(mods["relu"], nn.ReLU) or not isinstance(conv, nn.Conv2d):
        return None
    if bn is not None and (not isinstance(bn, nn.BatchNorm2d) or bn.training or bn.running_mean is None):
        return None
    if (conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.padding != (0, 0) or conv.dilation != (1, 1) or conv.groups != 1
            or conv.bias is not None or conv.padding_mode != "zeros"):
        return None
    return bn, conv


def _plain_bilinear(module) -> bool:
    """The module's upsampling function must BE F.interpolate(x, size, mode='bilinear', align_corners=False): checked by behaviour, once."""
    fn = getattr(module, "upsampling_method", None)
    if fn is None:
        return True
    ok = getattr(module, "_bc_spp_upsample_ok", None)
    if ok is None:
        t = torch.arange(24, dtype=torch.float32).reshape(1, 2, 3, 4) * 0.37
        try:
            ok = bool(torch.equal(fn(t, (5, 9)), F.interpolate(t, (5, 9), mode="bilinear", align_corners=False)))
        except Exception:
            ok = False
        module._bc_spp_upsample_ok = ok
    return ok


def match(module):
    """Structure of the reference's SpatialPyramidPooling (and of bc_workloads.swiftnet's restatement): (blocks, grids) or None."""
    if type(module).__name__ != "SpatialPyramidPooling" or module.training:
        return None
    spp, grids = getattr(module, "spp", None), getattr(module, "grids", None)
    if not isinstance(spp, nn.Sequential) or len(spp) < 3 or grids is None or getattr(module, "square_grid", False) or getattr(module, "fixed_size", None) is not None:
        return None
    blocks = [_block_parts(b) for b in spp]
    if any(b is None for b in blocks) or len(grids) < len(blocks) - 2 or not _plain_bilinear(module):
        return None
    levels, fuse = blocks[1:-1], blocks[-1]
    C, CO = levels[0][1].in_channels, levels[0][1].out_channels
    if any(c.in_channels != C or c.out_channels != CO or (bn is None) != (levels[0][0] is None) for bn, c in levels):
        return None
    if blocks[0][1].out_channels != C or fuse[1].in_channels != C + len(levels) * CO:
        return None
    return blocks, tuple(int(g) for g in grids[:len(levels)])


def _affine(bn):
    return fusion.batchnorm_affine(bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps)


def _params(module, blocks, be):
    """Folded BN vectors and packed weights of the level blocks and the last block, derived once per parameter state."""
    tensors = [t for bn, conv in blocks[1:] for t in ((bn.running_mean, bn.running_var, bn.weight, bn.bias, conv.weight) if bn is not None else (conv.weight,))
               if t is not None]
    key = tuple((t.data_ptr(), t._version) for t in tensors)
    cached = getattr(module, "_bc_spp_params", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    levels, (fbn, fconv) = blocks[1:-1], blocks[-1]
    with torch.no_grad():
        if levels[0][0] is not None:
            aff = [_affine(bn) for bn, _ in levels]
            lscale, lshift = torch.stack([a[0] for a in aff]).contiguous(), torch.stack([a[1] for a in aff]).contiguous()
        else:
            lscale = lshift = None
        lw = be.pack_spp_level_weights([c.weight for _, c in levels])
        fscale, fshift = (a.contiguous() for a in _affine(fbn)) if fbn is not None else (None, None)
        fw = be.pack_spp_fuse_weights(fconv.weight)
    out = (lscale, lshift, lw, fscale, fshift, fw)
    if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        module._bc_spp_params = (key, out)      # (memory of a graph's private pool must not outlive the capture as a cached constant)
    return out


def forward(module, x, like=None):
    """The module's forward on the dense TensorWrapper ``x`` through the two-launch route, or None (caller runs the generic route).
    ``like`` = the packed tensor the dense map was combined from: with one map per launch and exact tile counts the second launch then
    computes the executed tiles only and the result comes back PACKED (the caller must not re-pack it)."""
    if not fusion.SPP_FUSED or not fusion.ENABLED or torch.is_grad_enabled():
        return None
    be = get_backend()
    if not hasattr(be, "spp_levels"):
        return None
    m = match(module)
    if m is None:
        return None
    blocks, grids_cfg = m
    fconv = blocks[-1][1]
    CO, L, N = blocks[1][1].out_channels, len(blocks) - 2, fconv.out_channels
    raw = x._raw()
    if raw.dim() != 4 or raw.dtype not in getattr(be, "supports_fusion_dtypes", ()) or any(p.dtype != raw.dtype for p in (fconv.weight, blocks[1][1].weight)):
        return None
    x0 = module.spp[0](x)                                   # first block: the engine's own pointwise conv with the BN / ReLU as prologue
    x0 = x0._plain() if hasattr(x0, "_plain") else x0
    B, C, H, W = x0.shape
    # an NCHW model keeps NCHW maps (the kernels read and write channels-last ones, and what follows expects the layout the generic
    # route would have produced): the first block's output tells which kind of model this is
    ar = W / H
    grids = [(g, max(1, round(ar * g))) for g in grids_cfg]              # (reference: grid_size = (g, max(1, round(ar * g))))
    if not x0.is_contiguous(memory_format=torch.channels_last) or not be.spp_supported(x0[:1], CO, L, N, grids):
        return _generic_tail(module, x0, x)
    lscale, lshift, lw, fscale, fshift, fw = _params(module, blocks, be)
    lv = be.spp_levels(x0, lscale, lshift, lw, grids)          # (the whole batch in one launch each: the maps are independent)
    feats = like.get_features() if like is not None else None
    if (PACKED_RESULT and B == 1 and feats is not None and feats.dyn is None and feats.engine == "fused" and getattr(be, "spp_fuse_packed_ok", True)):
        mapping_exec = like.get_mapping_exec()
        bs = H // like.get_grid_idx().shape[2]
        out = be.spp_fuse(x0, lv, fscale, fshift, fw, grids, N, packed=(mapping_exec, bs))
        return type(x)._wrap_like(out, like, True)
    out = be.spp_fuse(x0, lv, fscale, fshift, fw, grids, N)
    return type(x)._wrap_like(out, x, False)


def _generic_tail(module, x0, like):
    """The rest of the module op by op on the already computed first block (shapes the two kernels do not cover)."""
    x0 = type(like)._wrap_like(x0, like, False)
    size = x0.shape[2:4]
    ar = size[1] / size[0]
    levels = [x0]
    for i in range(1, len(module.spp) - 1):
        g = module.grids[i - 1]
        pooled = F.adaptive_avg_pool2d(x0, (g, max(1, round(ar * g))))
        levels.append(F.interpolate(module.spp[i](pooled), size, mode="bilinear", align_corners=False))
    return module.spp[-1](torch.cat(levels, 1))
