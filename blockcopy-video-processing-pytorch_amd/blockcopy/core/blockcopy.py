"""Per-frame block-copy manager: policy -> pack -> base model on packed tiles -> combine -> policy update.

Same contract as the reference's ``BlockCopyModel`` / ``blockcopy_noblocks`` (core/blockcopy.py:7-122): same
constructor, ``forward`` / ``reset_temporal`` / ``load_state_dict``, same ``policy_meta`` keys, same per-clip
semantics (first frame of a clip executes every tile; ``num_exec == 0`` returns the cached output object)."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

import blockcopy

from ..utils.profiler import timings


class BlockCopyModel(nn.Module):
    """Wrap ``base_model`` so that only the tiles chosen by the policy are recomputed on each frame.

    settings: dict with the ``block_*`` keys of ``blockcopy.add_argparser_arguments``.
    """

    def __init__(self, base_model: nn.Module, settings: dict):
        super().__init__()
        self.is_blockcopy_manager = True   # marks the module that owns the temporal state
        self.base_model = base_model
        self.policy = blockcopy.build_policy_from_settings(settings)
        self.block_temporal_features = None
        self.train_interval = settings["block_train_interval"]
        self.block_size = settings["block_size"]
        # MI355X-first: replay the packed pipeline as a hipGraph (core/graphs.py): 1 = one graph per executed-tile count, 2 = ONE graph
        # for every count (launches sized for all tiles, the count read from the device: no wait for a device-side policy decision)
        self.use_graph = int(settings.get("block_graph", int(os.environ.get("BLOCKCOPY_GRAPH", "0"))))
        self.block_target = settings.get("block_target", None)
        self._graphed = {}
        self.reset_temporal()

    def load_state_dict(self, state_dict, strict: bool = True):
        """Checkpoints are those of the base model (reference: core/blockcopy.py:30-32)."""
        return self.base_model.load_state_dict(state_dict, strict=strict)

    def reset_temporal(self):
        """Forget all temporal state; call at the start of every clip."""
        self.clip_length = 0
        if self.block_temporal_features:
            self.block_temporal_features.clear()
        self.block_temporal_features = None
        self.policy_meta = {"inputs": None, "outputs": None, "outputs_prev": None}
        for gf in self._graphed.values():
            gf.reset()
        # NB: the reference also calls torch.cuda.empty_cache() here (:43).  Returning the whole caching
        # allocator to the driver once per clip forces hipFree/hipMalloc round trips on the next clip's first
        # frame; stale blocks are simply reused by the allocator, so it is not needed for correctness.

    def forward(self, inputs, **kwargs):
        return self._forward_blockcopy(inputs, **kwargs)

    def _forward_blockcopy(self, inputs, **kwargs):
        self.clip_length += 1

        self.policy_meta["inputs"] = inputs
        # tells a trainable policy whether this frame's decision will be trained on (it then needs an autograd graph)
        self.policy_meta["train_hint"] = self.clip_length % self.train_interval == 0
        if hasattr(self.policy, "wait_free"):
            # a policy that decides on the device need not tell the host its executed-tile count when the frame runs as the ONE
            # dynamic graph (every launch reads the count from the device); any other execution mode needs the number
            self.policy.wait_free = self.use_graph == 2 and blockcopy.core.tensorwrapper.ENGINE == "fused"
            if self.policy.wait_free:
                # ... and it can write its index tables straight into the captured frame's table buffer (same layout), once that exists
                gf = self._graphed.get((tuple(inputs.shape), inputs.dtype, inputs.device))
                if gf is not None:
                    self.policy_meta["tables_target"] = gf.tables[:2 * gf.n_total + 4]
        with timings.env("blockcopy/policy_forward", 3):
            # the policy writes the execution grid into policy_meta['grid'] (+ optional CPU mirror 'grid_host')
            self.policy_meta = self.policy(self.policy_meta)

        with timings.env("blockcopy/model", 3):
            x = blockcopy.to_tensorwrapper(inputs)
            num_exec = self.policy_meta["num_exec"]
            # (a count that only the device knows -- policy.LazyCount, dynamic graph -- is not asked for here: that would be the wait
            #  the mode exists to avoid; the graph copes with zero executed tiles, its output is then a copy of the previous map)
            if not getattr(num_exec, "lazy", False) and num_exec == 0:
                # nothing to execute: hand back the cached outputs
                self.policy_meta = self.policy_meta.copy()
                out = self.policy_meta["outputs"]
            elif self.use_graph and blockcopy.core.tensorwrapper.ENGINE == "fused":
                out = self._forward_graphed(inputs, **kwargs)
            else:
                self.policy_meta.pop("grid_tables", None)   # (eager engines build their tables from the host mirror)
                self.block_temporal_features = x.process_temporal_features(self.block_temporal_features)
                blocks = x.to_blocks(self.policy_meta["grid"], self.policy_meta.get("grid_host", None))
                # frame state = most recently executed pixels of every tile
                self.policy_meta["frame_state"] = blocks.combine_().to_tensor()
                out = self.base_model(blocks, **kwargs)
                out = out.combine().to_tensor()
                self.block_temporal_features.flush_deferred()

            self.policy_meta["outputs_prev"] = self.policy_meta["outputs"]
            self.policy_meta["outputs"] = out

        with timings.env("blockcopy/policy_optim", 3):
            if self.policy is not None:
                train_policy = self.clip_length % self.train_interval == 0
                self.policy_meta = self.policy.optim(self.policy_meta, train=train_policy)
        return out


@torch.no_grad()
def prewarm(self, inputs, counts=None, **kwargs):
    """Warm up and capture the packed pipeline for every executed-tile count the policy can produce, so that no
    MIOpen solver search or graph capture lands inside a clip.  ``counts`` defaults to the multiples of
    ``num_tiles / 16`` (the policies' quantisation step, policy.py ``quantize_number_exec_grid``).  Uses ``inputs`` only
    for its shape/dtype/device; temporal state is reset afterwards (call it between clips)."""
    from .graphs import GraphedFrame, WARM_RUNS

    assert self.use_graph, "prewarm() is for the hipGraph execution mode (settings['block_graph'] = 1 or 2)"
    key = (tuple(inputs.shape), inputs.dtype, inputs.device)
    gf = self._graphed.get(key)
    if gf is None:
        gf = self._graphed[key] = GraphedFrame(inputs, self.block_size, dynamic=self.use_graph == 2, plan_fraction=self.block_target)
    total = gf.n_total
    if gf.dynamic:
        counts = [total]        # one graph serves every count: warm and capture it on the all-active frame
    if counts is None:
        step = max(1, total // 16)
        counts = sorted(set(list(range(step, total + 1, step)) + [total]), reverse=True)
    assert counts[0] == total or gf.state.rings, "the first warmed count must be the all-active frame"
    grid_dev = torch.ones(gf.grid_shape, dtype=torch.bool, device=inputs.device)
    for n in counts:
        host = torch.zeros(total, dtype=torch.bool)
        host[:n] = True
        for _ in range(WARM_RUNS + 1):           # eager warm run(s), then capture + first replay
            gf.prev_out = None if n == total else gf.prev_out
            gf.upload(inputs, host.view(gf.grid_shape))
            gf.finish(gf.run(self.base_model, n, grid_dev, **kwargs))
    torch.cuda.synchronize(inputs.device)
    self.reset_temporal()
    return counts


BlockCopyModel.prewarm = prewarm


def _forward_graphed(self, inputs, **kwargs):
    from .graphs import GraphedFrame

    key = (tuple(inputs.shape), inputs.dtype, inputs.device)
    gf = self._graphed.get(key)
    if gf is None:
        gf = self._graphed[key] = GraphedFrame(inputs, self.block_size, dynamic=self.use_graph == 2, plan_fraction=self.block_target)
    grid = self.policy_meta["grid"]
    dev_tables = self.policy_meta.pop("grid_tables", None)
    if dev_tables is not None:
        n_exec = gf.upload_tables(inputs, *dev_tables)      # tables built by the device policy step: no host table work
    else:
        grid_host = self.policy_meta.get("grid_host", None)
        if grid_host is None:
            grid_host = grid.to("cpu")   # device-only grid: the one D->H sync of the frame
        n_exec = gf.upload(inputs, grid_host)
    out_blocks = gf.run(self.base_model, n_exec, grid, **kwargs)
    self.policy_meta["frame_state"] = gf.frame_state
    return gf.finish(out_blocks)


BlockCopyModel._forward_graphed = _forward_graphed


def blockcopy_noblocks(func):
    """Decorator for a Module ``forward`` that needs the dense map (e.g. global pooling): combine the packed
    input in place, run the module densely, re-pack the result with the same grid.

        class GlobalContext(nn.Module):
            @blockcopy_noblocks
            def forward(self, x): ...
    """

    def noblocks(self, x, *args):
        packed = isinstance(x, blockcopy.TensorWrapper)
        if packed:
            like = x
            x = x.combine_()
            # fused engine: the dense map stays a (non-block) TensorWrapper, so eval batch-norm / ReLU inside the dense
            # module are recorded and folded into ONE launch like on packed tensors (e.g. the BN -> ReLU -> 1x1 conv
            # blocks of SwiftNet's pyramid pooling: stock BN + two layout copies + ReLU = 37 us on a 4 MB map, fused 5 us)
            if not x.fuses_dense_ops:
                x = x.to_tensor()
        fast = None
        if packed and isinstance(x, blockcopy.TensorWrapper) and not args:
            from . import spp_fused            # the reference's pyramid-pooling module: two launches instead of 15 (csrc/spp.inc)

            fast = spp_fused.forward(self, x, like)
        x = fast if fast is not None else func(self, x)
        if packed and not (fast is not None and blockcopy.is_block(fast)):      # (the fast route may hand the executed tiles back packed)
            x = blockcopy.to_tensorwrapper(x).to_blocks_like(like)
        return x

    return noblocks
