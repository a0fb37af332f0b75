"""The three packed<->dense copy ops behind the reference's autograd-Function names
(SplitFunction / CombineFunction / TransferFunction, utils/block_funcs.py:10-237), bound to the gfx950
kernels of libblockcopy_hip.so.  Inference only: ``backward`` raises, as in the reference (:52,:127,:196)."""
from __future__ import annotations

from torch.autograd import Function

from ..backend import get_backend
from .profiler import timings


class SplitFunction(Function):
    @staticmethod
    def forward(ctx, blocks, image, mapping_exec, grid_idx, dyn=None):
        """Copy the executed tiles of ``image`` (N,C,H,W) into ``blocks`` (n_exec,C,bs,bs).  (``dyn``: device-side executed-tile
        count of a ceiling-sized launch, HipBackend._arm; not part of the reference's signature.)"""
        return get_backend().split(blocks, image, mapping_exec, grid_idx) if dyn is None else get_backend().split(blocks, image, mapping_exec, grid_idx, dyn=dyn)

    @staticmethod
    def backward(ctx, grad_x):
        raise NotImplementedError()


class CombineFunction(Function):
    @staticmethod
    def forward(ctx, blocks, out, grid_idx, mapping_exec, dyn=None):
        """Scatter ``blocks`` into the dense map ``out`` in place (other tiles keep the previous frame's values)."""
        with timings.env("block/combine_kernel", 20):
            be = get_backend()
            return be.combine(blocks, out, grid_idx, mapping_exec) if dyn is None else be.combine(blocks, out, grid_idx, mapping_exec, dyn=dyn)

    @staticmethod
    def backward(ctx, grad_x):
        raise NotImplementedError()


class TransferFunction(Function):
    @staticmethod
    def forward(ctx, data_transfer, prev_computed, prev_transfer, grid_idx_prev, transfer_map_prev, padding):
        """Fill the border ring of ``data_transfer`` from the previous frame's computed / transferred tiles."""
        with timings.env("block/transfer_kernel", 20):
            return get_backend().transfer(data_transfer, prev_computed, prev_transfer, grid_idx_prev, transfer_map_prev, padding)

    @staticmethod
    def backward(ctx, grad_data_tansfer):
        raise NotImplementedError()


class CombineCopyFunction(Function):
    """MI355X-first: fused scatter + copy (replaces ``prev.clone()`` + CombineFunction, core/tensorwrapper.py:421-433)."""

    @staticmethod
    def forward(ctx, blocks, prev, out, grid_idx):
        with timings.env("block/combine_copy_kernel", 20):
            return get_backend().combine_copy(blocks, prev, out, grid_idx)

    @staticmethod
    def backward(ctx, grad_x):
        raise NotImplementedError()
