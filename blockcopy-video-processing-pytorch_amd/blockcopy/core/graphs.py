"""hipGraph execution of the per-frame packed pipeline.

Per frame the eager engine issues ~160 kernel launches through ~100 ``__torch_function__`` dispatches; at SwiftNet-RN18
/ 1024x2048 that is 3 ms of host time for 2.8 ms of GPU time, and the small kernels leave launch gaps.  The packed
pipeline is shape-static once the number of executed tiles is fixed, so it is captured ONCE per executed-tile count
(the policies quantise that count to a handful of values, policy.py ``quantize_number_exec_grid``) into a hipGraph over
buffers with fixed addresses:

    static input frame -> gather -> in-place scatter (frame_state) -> base model on packed tiles [halo gathers over
    persistent ring caches, MIOpen convs, per-tile resampling, dense SPP ...] -> packed output tiles

Per frame the host then does: policy -> index tables (C loop, pinned) -> one H->D copy into the static table buffer ->
one D->D copy of the frame -> ``graph.replay()`` -> the final fused scatter+copy into a fresh output tensor (eager, so
callers may keep every frame's output, as with the reference).  No Python runs per layer and nothing synchronises.

Reference behaviour being reproduced: BlockCopyModel._forward_blockcopy, core/blockcopy.py:62-79.
"""
from __future__ import annotations

import numpy as np
import torch

from ..backend import empty_like_layout, get_backend, pinned_ring
from .tensorwrapper import BlockFeatures, PersistentState, TensorWrapper, _NoDispatch

WARM_RUNS = 1   # eager runs of a new executed-tile count before it is captured (MIOpen solver search, lazy module loads)


class _Bucket:
    __slots__ = ("warm", "graph", "out_blocks")

    def __init__(self):
        self.warm = 0
        self.graph = None
        self.out_blocks = None


class GraphedFrame:
    """Static buffers + captured graphs for one (input shape, dtype)."""

    def __init__(self, inputs: torch.Tensor, block_size: int):
        N, C, H, W = inputs.shape
        assert H % block_size == 0 and W % block_size == 0
        self.block_size = block_size
        self.grid_shape = (N, 1, H // block_size, W // block_size)
        self.n_total = N * (H // block_size) * (W // block_size)
        self.device = inputs.device
        self.static_in = torch.empty_like(inputs, memory_format=torch.contiguous_format)
        self.tables = torch.zeros(2 * self.n_total, dtype=torch.int32, device=self.device)   # [grid_idx | mapping_exec]
        self.grid_idx = self.tables[:self.n_total].view(self.grid_shape)
        self.state = PersistentState()
        self.buckets = {}
        self.pool = None
        self.prev_out = None      # previous frame's dense output (None at the start of a clip)
        self.frame_state = None

    # ------------------------------------------------------------------ per-frame host work
    def upload(self, inputs: torch.Tensor, grid_host: torch.Tensor) -> int:
        g8 = grid_host.to(torch.bool).contiguous().numpy().view(np.uint8).reshape(-1)
        assert g8.size == self.n_total
        ring = pinned_ring(2 * self.n_total, torch.int32, self.device.type == "cuda")     # reused pinned staging (no per-frame page-locking)
        staging = ring.next()
        st = staging.numpy()
        n_exec = get_backend().grid_tables_host(g8, st[:self.n_total], st[self.n_total:], None, None)
        self.tables.copy_(staging, non_blocking=True)
        ring.uploaded()
        if inputs.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(inputs, non_blocking=True)
        return n_exec

    def upload_tables(self, inputs: torch.Tensor, tables: torch.Tensor, n_exec: int) -> int:
        """Per-frame work when the policy already built the index tables on the device (bc_policy_step): two D->D copies."""
        assert tables.numel() == 2 * self.n_total and tables.dtype == torch.int32
        self.tables.copy_(tables, non_blocking=True)
        if inputs.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(inputs, non_blocking=True)
        return int(n_exec)

    # ------------------------------------------------------------------ the capturable body
    def body(self, base_model, n_exec: int, grid: torch.Tensor, **kwargs):
        feats = BlockFeatures(self.device, engine="fused")
        feats.persistent = self.state
        self.state.rewind()
        feats._grid = grid
        feats._grid_idx = self.grid_idx
        feats._mapping_exec = self.tables[self.n_total:self.n_total + n_exec]
        feats.n_exec, feats.n_total = n_exec, self.n_total
        with _NoDispatch():
            x = self.static_in.as_subclass(TensorWrapper)
        x._init_metadata()
        x._features = feats
        blocks = x._split(self.block_size)
        frame_state = blocks.combine_()._plain()
        out = base_model(blocks, **kwargs)
        if isinstance(out, TensorWrapper) and out.is_blocks:
            return out._plain(), frame_state          # packed output tiles: combined eagerly by finish()
        # models that combine inside (e.g. the CSP head): dense maps living in the persistent state / graph pool
        from .tensorwrapper import to_tensor
        return to_tensor(out), frame_state

    def run(self, base_model, n_exec: int, grid: torch.Tensor, **kwargs):
        """Outputs of ``base_model(packed tiles)`` for this frame -- packed output tiles, or whatever dense structure
        the model returns -- eager while warming up, graph replay afterwards (``base_model`` may be any callable)."""
        if self.prev_out is None:
            assert n_exec == self.n_total, "No previous features known, first run should execute all blocks!"
        b = self.buckets.setdefault(n_exec, _Bucket())
        if b.graph is None and (b.warm < WARM_RUNS or self.device.type != "cuda"):   # (no graphs off-GPU: test hook only)
            b.warm += 1
            out_blocks, self.frame_state = self.body(base_model, n_exec, grid, **kwargs)
            return out_blocks
        if b.graph is None:
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.pool):
                b.out_blocks, self.frame_state = self.body(base_model, n_exec, grid, **kwargs)
            b.graph = g
            self.state.frozen = True
        b.graph.replay()
        return b.out_blocks

    def finish(self, out_blocks: torch.Tensor) -> torch.Tensor:
        """Final out-of-place combine: fused scatter+copy against the previous frame's output (eager: every frame's
        output is a fresh tensor the caller may keep)."""
        be = get_backend()
        n_exec, C, bs, _ = out_blocks.shape
        N, _, GH, GW = self.grid_shape
        out = empty_like_layout((N, C, GH * bs, GW * bs), out_blocks)
        if self.prev_out is None:
            be.combine(out_blocks, out, self.grid_idx, self.tables[self.n_total:self.n_total + n_exec])
        else:
            be.combine_copy(out_blocks, self.prev_out, out, self.grid_idx)
        self.prev_out = out
        return out

    def reset(self):
        self.prev_out = None
