"""Halo padding of the packed batch (reference: ``pad`` / BlockPadFunction, utils/blockpad.py:14-156) on the
gfx950 halo-gather kernel, plus the ring-cache form used by the fused engine."""
from __future__ import annotations

import warnings

from torch.autograd import Function

from ..backend import get_backend
from .profiler import timings


def _warn_tiny(blocksize: int):
    if blocksize <= 2:
        warnings.warn(f"Block size of 2 or smaller can be inefficient! Got size: {blocksize}")


def pad(features, transfer, grid_idx, exec_map, pad=1):
    """(n_exec,C,bs,bs) -> (n_exec,C,bs+2p,bs+2p) with halo from neighbouring tiles / previous frame / zeros."""
    return BlockPadFunction.apply(features, transfer, grid_idx, exec_map, pad)


def pad_ring(features, ring, grid_idx, exec_map, pad=1, prologue=None, dyn=None):
    """Same result as transfer + pad of the reference, over a persistent per-layer ring cache.
    ``prologue`` = (scale, shift, relu) fuses a pending per-channel affine + ReLU into the gather; ``dyn`` = device-side executed-tile
    count of a ceiling-sized launch (HipBackend._arm)."""
    if dyn is None:
        return BlockPadRingFunction.apply(features, ring, grid_idx, exec_map, pad, prologue)
    return BlockPadRingFunction.apply(features, ring, grid_idx, exec_map, pad, prologue, dyn)


class BlockPadFunction(Function):
    @staticmethod
    def forward(ctx, data_exec, data_transfer, grid_idx, mapping_exec, pad):
        _warn_tiny(data_exec.shape[2])
        with timings.env("block/pad_kernel", 20):
            return get_backend().pad(data_exec, data_transfer, grid_idx, mapping_exec, pad)

    @staticmethod
    def backward(ctx, grad_x):
        raise NotImplementedError("Backward not implemented for BlockPad")


class BlockPadRingFunction(Function):
    @staticmethod
    def forward(ctx, data_exec, ring, grid_idx, mapping_exec, pad, prologue=None, dyn=None):
        _warn_tiny(data_exec.shape[2])
        with timings.env("block/pad_kernel", 20):
            if dyn is None:
                return get_backend().pad_ring(data_exec, ring, grid_idx, mapping_exec, pad, prologue)
            return get_backend().pad_ring(data_exec, ring, grid_idx, mapping_exec, pad, prologue, dyn=dyn)

    @staticmethod
    def backward(ctx, grad_x):
        raise NotImplementedError("Backward not implemented for BlockPad")
