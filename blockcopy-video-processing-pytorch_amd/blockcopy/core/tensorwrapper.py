"""Block-sparse tensor representation with temporal feature propagation.

Public surface mirrors the reference's ``core/tensorwrapper.py`` (``TensorWrapper``, ``BlockFeatures``,
``to_tensorwrapper / to_tensor / is_block / is_tensorwrapper``) so model code written for the reference
(`blockcopy.to_tensor(x)`, ``@blockcopy_noblocks``, ``x.combine_()``, ...) runs unchanged.  What differs is
how the path is executed on MI355X:

* index tables (``grid_idx``, ``mapping_exec``) are built by one C loop into a pinned staging buffer and
  uploaded with a single asynchronous H->D copy; when the policy supplies a host mirror of its grid there is
  **no** device->host synchronisation in the frame (the reference does two syncs + three uploads + a
  GPU-tensor-by-CPU-mask index per frame, core/tensorwrapper.py:150-178);
* ``ENGINE == "fused"`` (default): every padded layer owns a persistent **ring cache**; one ``pad_ring``
  launch builds the halo-padded packed batch and refreshes the executed tiles' rings, so the reference's
  per-layer ``transfer`` launch, its ``(computed, transfer)`` tensor pair and their FIFO disappear
  (reference: _transfer_from_prev :445-476 + store_features :180-189 + pad);
* non-in-place ``combine`` is ONE fused scatter+copy launch instead of ``clone()`` + scatter (:421-433);
* ``ENGINE == "reference"`` keeps the reference's exact decomposition (split / transfer / pad / combine with
  the FIFO of tensor pairs) on the same HIP library -- used for op-by-op parity checks.

``__torch_function__`` is a classmethod (the instance-method form is deprecated in current PyTorch) and all
handler bodies run with subclass dispatch disabled, so attribute reads such as ``x.shape`` inside the engine
cost nothing extra.
"""
from __future__ import annotations

import os
import warnings
from collections import deque
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np
import torch

from . import fusion
from ..backend import dense_layout, empty_like_layout, get_backend, is_nhwc, pinned_ring
from ..utils.block_funcs import CombineCopyFunction, CombineFunction, SplitFunction, TransferFunction
from ..utils.blockpad import pad, pad_ring
from ..utils.profiler import timings

VERBOSE = False                # debugging flag: print verbose statements
BLOCKPAD_WITH_ZEROES = False   # debugging flag: zero halo instead of block padding (reference :14,:536)
ENGINE = os.environ.get("BLOCKCOPY_ENGINE", "fused")   # "fused" | "reference"

_NoDispatch = torch._C.DisableTorchFunctionSubclass


def set_engine(name: str) -> str:
    """Select the execution engine ("fused" or "reference"); returns the previous one."""
    global ENGINE
    assert name in ("fused", "reference"), name
    prev, ENGINE = ENGINE, name
    return prev


def is_tensorwrapper(x) -> bool:
    """True if x is a TensorWrapper."""
    return isinstance(x, TensorWrapper)


def is_block(x) -> bool:
    """True if x is a TensorWrapper in packed (blocks) representation."""
    return isinstance(x, TensorWrapper) and x.is_blocks


def to_tensorwrapper(x: torch.Tensor) -> "TensorWrapper":
    """View a tensor as a TensorWrapper (no copy)."""
    # reference asserts x.is_cuda (core/tensorwrapper.py:35); here the GPU requirement is enforced by the HIP
    # backend itself, which is the only backend the package can load on its own.
    assert x.is_cuda or get_backend().name != "hip", "x must be on the GPU!"
    with _NoDispatch():
        return x.as_subclass(TensorWrapper)


_CONV2D = torch.nn.functional.conv2d
_S2_PLAN_OK = {}      # (plan, pixels, tile size, cin, cout, dtype) -> does this stride-2 pointwise decomposition cover that geometry


class DenseMap(torch.Tensor):
    """What ``to_tensor`` hands out for a wide channels-last map under the fused engine: an ordinary dense tensor in every respect
    but one -- a 3x3 / stride 1 / padding 1 conv to at most 4 output channels on it (the prediction convs a detector head applies
    right after ``blockcopy.to_tensor``, reference Pedestron/mmdet/models/anchor_heads/csp_head.py:139-151) runs in the bandwidth
    kernel ``bc_pred3x3_nhwc`` instead of the conv library (200-410 us -> 20-35 us per (1,256,256,512) map).  Every other op
    sees, and returns, plain tensors."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is _CONV2D and fusion.PRED_KERNEL:
            got = _dense_pred_conv(args, kwargs)
            if got is not None:
                return got
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def _dense_pred_conv(args, kwargs):
    """conv2d(DenseMap, weight (<= 4, Cin, 3, 3), bias, stride 1, padding 1) through bc_pred3x3_nhwc, or None."""
    cv = {k: kwargs.get(k, args[i] if len(args) > i else d) for k, i, d in
          (("input", 0, None), ("weight", 1, None), ("bias", 2, None), ("stride", 3, 1), ("padding", 4, 0), ("dilation", 5, 1), ("groups", 6, 1))}
    x, weight, bias = cv["input"], cv["weight"], cv["bias"]
    if not (isinstance(x, DenseMap) and isinstance(weight, torch.Tensor) and not isinstance(weight, DenseMap)) or torch.is_grad_enabled() and (
            x.requires_grad or weight.requires_grad):
        return None
    be = get_backend()
    with torch._C.DisableTorchFunctionSubclass():
        xt = x.as_subclass(torch.Tensor)
        if not (hasattr(be, "pred3x3") and be.pred3x3_supported(xt, weight, cv["stride"], cv["padding"], cv["dilation"], cv["groups"])):
            return None
        wpk = fusion.packed_conv3x3_weight(weight, be.pack_pred3x3_weights)        # cached per parameter object
        b32 = None if bias is None else fusion.packed_conv3x3_weight(bias, lambda b: b.detach().float().contiguous())
        return be.pred3x3(xt, wpk, b32, weight.shape[0])


def to_tensor(x):
    """TensorWrapper (or list / tuple / dict of them) -> torch.Tensor; packed tensors are combined first."""
    if isinstance(x, TensorWrapper):
        return x.to_tensor()
    if isinstance(x, list):
        return [to_tensor(z) for z in x]
    if isinstance(x, tuple):
        return tuple(to_tensor(z) for z in x)
    if isinstance(x, dict):
        return {k: to_tensor(v) for k, v in x.items()}
    return x


# Op classes of the reference's routing table (core/tensorwrapper.py:69-105).
OPS = {
    # ops with a spatial padding argument: the zero padding is replaced by a halo gathered from neighbours
    "PADDED": {"conv2d", "max_pool2d", "avg_pool2d", "lp_pool2d", "fractional_max_pool2d"},
    # resampling ops: executed per tile, without halo
    "INTERPOLATE": {"interpolate", "upsample_bilinear"},
    # ops whose statistics run over the batch axis, which here is the tile axis
    "BATCHED": {"group_norm"},
    # ops that need the whole dense tensor
    "INCOMPATIBLE": {"adaptive_avg_pool2d", "adaptive_max_pool2d", "linear", "flip", "unsqueeze", "reshape", "view"},
    # reductions that are only meaningful over the channel axis
    "CHANNELONLY": {"mean", "sum", "max", "min,", "std", "var", "argmax", "count_nonzero", "nonzero"},
    "WARNING": {""},
}
OPS_SPECIAL = set().union(*OPS.values())

# positional index of `padding` for ops that reach __torch_function__ with positional arguments
_PADDING_POS = {"conv2d": 4, "avg_pool2d": 3, "max_pool2d": 3}


class PersistentState:
    """Temporal state with fixed HBM addresses, for the graph-captured frame pipeline (core/graphs.py).

    Holds, in call order, one ring cache per padded op and one dense map per combine call site.  Unlike the
    per-frame FIFO hand-over of the eager engine, these buffers survive ``reset_temporal()`` (an all-active first
    frame overwrites them completely), so a hipGraph captured over them stays valid for the life of the model."""

    def __init__(self):
        self.rings = []
        self.maps = []
        self.ring_pos = 0
        self.map_pos = 0
        self.frozen = False   # set once a graph has been captured: no new buffers may appear

    def rewind(self):
        self.ring_pos = 0
        self.map_pos = 0

    def _next(self, store, pos, shape, dtype, device, what, nhwc=False):
        if pos == len(store):
            assert not self.frozen, f"a new {what} appeared after graph capture; the model must run the same op sequence every frame"
            fmt = torch.channels_last if nhwc and len(shape) == 4 else torch.contiguous_format
            store.append(torch.empty(shape, dtype=dtype, device=device, memory_format=fmt))
        buf = store[pos]
        if tuple(buf.shape) != tuple(shape) or buf.dtype != dtype or (len(shape) == 4 and shape[1] > 1 and is_nhwc(buf) != bool(nhwc)):
            raise AssertionError(f"{what} #{pos}: expected {tuple(buf.shape)}/{buf.dtype}, got {tuple(shape)}/{dtype}; "
                                 "the model must run the same op sequence every frame")
        return buf

    def next_ring(self, shape, dtype, device):
        self.ring_pos += 1
        return self._next(self.rings, self.ring_pos - 1, shape, dtype, device, "padded op")

    def next_map(self, shape, dtype, device, nhwc=False):
        self.map_pos += 1
        return self._next(self.maps, self.map_pos - 1, shape, dtype, device, "combine call site", nhwc)


class BlockFeatures:
    """Per-frame block state: the execution grid with its index tables, plus the temporal feature store.

    fused engine     : ``rings`` -- one persistent (n_total,C,4*p*bs) compact ring cache per padded op, in call order,
                       handed from frame to frame;
    reference engine : FIFO of (computed, transfer, padding) per padded op as in the reference (:131-232);
    both             : FIFO of combined dense maps (``store_features_full``).
    """

    def __init__(self, device, engine: str = None):
        self.device = torch.device(device)
        self.engine = engine or ENGINE
        self._grid = None            # bool (N,1,GH,GW): tiles executed this frame
        self._grid_idx = None        # int32 (N,1,GH,GW): packed row of executed tiles; -n_total+k for the k-th skipped tile
        self._mapping_exec = None    # int32 [n_exec]: flat grid position of each packed row
        self._transfer_idx = None    # int32 [n_transfer]: previous frame's grid_idx at this frame's skipped tiles (reference engine)
        self._grid_idx_host = None   # numpy mirror of _grid_idx (host copy kept for the next frame's transfer_idx)
        self.n_exec = 0
        self.n_total = 0

        self._features_computed = deque()
        self._features_transfer = deque()
        self._features_full = deque()

        self.rings = []              # fused engine: persistent ring caches (shared list object across frames)
        self._ring_pos = 0
        self._pad_memo = None        # (source, padding, prologue key, padded): consecutive padded ops on the SAME tensor share one gather
        self._deferred = []          # deferred producers of this frame whose launch also refreshes a ring cache (fusion.Pending.defer_conv)
        self.persistent = None       # PersistentState when running as a graph-capturable body (core/graphs.py)
        # (n_exec_dev: device int32[1], ceiling) when the body is sized for a ceiling and the executed-tile count is read from the
        # device by every packed-tensor launch (core/graphs.py dynamic mode; include/blockcopy_hip.h bc_dyn_set); None = exact shapes
        self.dyn = None
        # dynamic mode: the executed-tile count the conv plan table is asked about (the kernel FORM of a layer is chosen for the count
        # the policy is expected to produce, the launch is sized for the ceiling); None = the actual count
        self.plan_n_exec = None

    # ------------------------------------------------------------------ grid -> index tables
    def _process_grid(self, grid: torch.Tensor, meta_prev: "BlockFeatures" = None, grid_host: torch.Tensor = None) -> None:
        with timings.env("tensorwrapper/process_grid", 10):
            assert grid.dim() == 4 and grid.shape[1] == 1, "grid must be (N,1,GH,GW)"
            if grid_host is None:
                # device-resident grid without host mirror: one D->H copy (the reference's path, :158)
                grid_host = grid.to("cpu", dtype=torch.bool)
            if grid_host.dtype != torch.bool:
                grid_host = grid_host.to(torch.bool)
            g8 = grid_host.contiguous().numpy().view(np.uint8).reshape(-1)
            n_total = g8.size
            on_gpu = self.device.type == "cuda"
            if on_gpu:
                ring = pinned_ring(3 * n_total, torch.int32, True)        # reused pinned staging (no per-frame page-locking)
                staging = ring.next()
            else:
                staging = torch.empty(3 * n_total, dtype=torch.int32)    # CPU tier: the tables ARE this tensor, so it must be fresh
            st = staging.numpy()
            want_transfer = meta_prev is not None and self.engine == "reference"
            n_exec = get_backend().grid_tables_host(
                g8, st[:n_total], st[n_total:2 * n_total],
                meta_prev._grid_idx_host if want_transfer else None,
                st[2 * n_total:] if want_transfer else None)
            if meta_prev is None:
                assert n_exec == n_total, "No previous features known, first run should execute all blocks!"
            else:
                assert meta_prev.n_total == n_total, "grid size changed inside a clip"
            dev = staging.to(self.device, non_blocking=True) if on_gpu else staging
            if on_gpu:
                ring.uploaded()
            self._grid = grid.to(self.device, dtype=torch.bool)
            self._grid_idx = dev[:n_total].view(grid.shape)
            self._mapping_exec = dev[n_total:n_total + n_exec]
            if want_transfer:
                self._transfer_idx = dev[2 * n_total:3 * n_total - n_exec]
            self._grid_idx_host = st[:n_total].copy() if on_gpu else st[:n_total]   # (the pinned staging is reused: keep an own mirror)
            self._staging = staging
            self.n_exec, self.n_total = n_exec, n_total
            if meta_prev is not None:
                self.rings = meta_prev.rings   # ring caches persist across frames

    # ------------------------------------------------------------------ fused engine: ring caches
    def next_ring(self, data: torch.Tensor, padding: int) -> torch.Tensor:
        """Ring cache of the padded op being executed (ops must come in the same order every frame).
        Compact layout: per grid position and channel [top p rows | bottom p rows | left p cols | right p cols]."""
        _, C, bs, _ = data.shape
        shape = (self.n_total, C, 4 * padding * bs)
        if self.persistent is not None:
            return self.persistent.next_ring(shape, data.dtype, data.device)
        k = self._ring_pos
        self._ring_pos += 1
        if k == len(self.rings):
            assert self.n_exec == self.n_total, "a new padded layer appeared after the first frame of the clip"
            self.rings.append(torch.empty(shape, dtype=data.dtype, device=data.device))
        ring = self.rings[k]
        if tuple(ring.shape) != shape or ring.dtype != data.dtype:
            raise AssertionError(f"padded op #{k}: expected tiles {tuple(ring.shape)}/{ring.dtype}, got {shape}/{data.dtype}; "
                                 "the model must run the same op sequence every frame")
        return ring

    def flush_deferred(self) -> int:
        """End of the frame body: run every deferred fused conv nobody asked the value of.  Its output is dropped, but its
        launch is what refreshes the layer's ring cache for the executed tiles -- without it a later frame that skips those
        tiles would gather this layer's halo from stale records.  Returns the number of launches made."""
        n = 0
        for launch, kw, state in self._deferred:
            if not state["launched"]:
                src, version = state["guard"]
                if src._version != version:
                    del self._deferred[:]
                    raise RuntimeError("blockcopy lazy fusion: the input of a deferred conv was modified in place before the end of the "
                                       "frame and its result was never consumed; set BLOCKCOPY_DEFER_CONV=0 / BLOCKCOPY_FUSE=0")
                launch(epilogue=None, **kw)
                state["launched"] = True
                n += 1
        del self._deferred[:]
        return n

    # ------------------------------------------------------------------ reference engine: FIFO of tensor pairs
    def store_features(self, data_computed, data_transfer, padding: int = -1) -> None:
        assert data_computed.shape[1:] == data_transfer.shape[1:], \
            f"Number of channels must be equal, got {(data_computed.shape, data_transfer.shape)}"
        self._features_computed.append((data_computed, padding))
        self._features_transfer.append((data_transfer, padding))

    def get_features(self):
        if len(self._features_computed) == 0:
            raise AssertionError("No computed features to pop from stack, something seems wrong in the model.")
        return self._features_computed.popleft(), self._features_transfer.popleft()

    # ------------------------------------------------------------------ combined dense maps
    def store_features_full(self, data) -> None:
        self._features_full.append(data)

    def get_features_full(self):
        if len(self._features_full) == 0:
            raise AssertionError("No combined features to pop from stack, something seems wrong in the model.")
        return self._features_full.popleft()

    def clear(self):
        self._features_computed.clear()
        self._features_transfer.clear()
        self._features_full.clear()
        self.rings = []


_FUSABLE = {"relu", "relu_", "batch_norm", "add", "add_", "__add__", "__iadd__"}


def _materialize_args(items):
    """Execute pending work of every TensorWrapper among (possibly nested) arguments."""
    for a in items:
        if isinstance(a, TensorWrapper):
            if a._pending is not None:
                a._materialize()
        elif isinstance(a, (list, tuple)):
            _materialize_args(a)


def _find_wrapper(items):
    """First initialised TensorWrapper among (possibly nested) arguments."""
    found = None
    for a in items:
        if isinstance(a, TensorWrapper):
            if a.is_init:
                return a
            found = found or a
        elif isinstance(a, (list, tuple)):
            r = _find_wrapper(a)
            if r is not None:
                if r.is_init:
                    return r
                found = found or r
    return found


class TensorWrapper(torch.Tensor):
    """torch.Tensor subclass carrying block metadata; packed tensors have shape (n_exec, C, bs, bs)."""

    is_init = False
    _pending = None    # fusion.Pending: elementwise work recorded but not yet executed (packed tensors only)

    # ------------------------------------------------------------------ metadata
    def _init_metadata(self, other=None):
        if not self.is_init:
            if other is None:
                self._is_blocks = False
                self._features = None
                self._features_prev = None
            else:
                self._is_blocks = other._is_blocks
                self._features = other._features
                self._features_prev = other._features_prev
            self.is_init = True
        return self

    def process_temporal_features(self, features_prev: BlockFeatures = None) -> BlockFeatures:
        """Start a new frame: attach a fresh BlockFeatures, remember the previous frame's."""
        self._init_metadata()
        self._features_prev = features_prev
        with _NoDispatch():
            self._features = BlockFeatures(device=self.device)
        return self._features

    @property
    def data_shape(self) -> torch.Size:
        return self.shape

    @property
    def is_blocks(self) -> bool:
        return self._is_blocks

    @property
    def block_size(self) -> int:
        return self.data_shape[-1] if self.is_blocks else -1

    def get_grid(self):
        return self._features._grid

    def get_grid_idx(self):
        return self._features._grid_idx

    def get_mapping_exec(self):
        return self._features._mapping_exec

    def get_features(self) -> BlockFeatures:
        return self._features

    def _dyn(self):
        """Device-side executed-tile count for launches on THIS tensor: the frame's (n_exec_dev, ceiling) for a packed tensor, None
        for dense maps (their size does not depend on the count) and for frames with exact shapes."""
        f = self._features
        return f.dyn if (f is not None and self._is_blocks) else None

    def _dyn_kw(self) -> dict:
        """``{"dyn": ...}`` for backend calls on this tensor, or ``{}`` (checker backends and exact-shape frames never see the keyword)."""
        d = self._dyn()
        return {} if d is None else {"dyn": d}

    def _plain(self) -> torch.Tensor:
        """Plain-tensor view of the VALUE (pending elementwise work is executed first)."""
        if self._pending is not None:
            self._materialize()
        with _NoDispatch():
            return self.as_subclass(torch.Tensor)

    def _raw(self) -> torch.Tensor:
        """Plain-tensor view of the stored data, ignoring pending work."""
        with _NoDispatch():
            return self.as_subclass(torch.Tensor)

    def _materialize(self) -> "TensorWrapper":
        """Execute the pending affine/add/ReLU in ONE fused pass into a new buffer and rebind this tensor to it
        (the raw buffer is left intact: other tensors may still refer to it)."""
        P = self._pending
        if P is None:
            return self
        self._pending = None
        with _NoDispatch():
            raw = dense_layout(self.as_subclass(torch.Tensor))
            add = P.checked_add()
            if add is not None:
                add = dense_layout(add)
            be = get_backend()
            P.check_source()
            if P.conv is not None:        # deferred fused halo+conv: launch now, the rest of the record is its epilogue
                launch, kw, state = P.conv
                plain = P.scale is None and P.shift is None and add is None and not P.relu
                if P.up is not None:      # + bilinear(coarser map) in the epilogue of the pointwise conv
                    usrc, uH, uW, ualign, urh, urw = P.up
                    kw = dict(kw, upsample=(usrc, uH, ualign, urh, urw))
                out = launch(epilogue=None if plain else (P.scale, P.shift, add, P.relu), **kw)
                state["launched"] = True
            elif P.interp is not None:    # deferred interpolation: resample now, the rest of the record is its epilogue
                src, H, W, align, rh, rw = P.interp
                plain = P.scale is None and P.shift is None and add is None and not P.relu
                out = be.interp_bilinear(src, H, W, align, rh, rw, None if plain else (P.scale, P.shift, add, P.relu), **self._dyn_kw())
            elif raw.dtype in getattr(be, "supports_fusion_dtypes", ()):
                out = be.affine_act(raw, P.scale, P.shift, add, P.relu, **self._dyn_kw())
            else:   # exotic dtype: the same arithmetic with stock ops
                out = raw.float()
                if P.scale is not None:
                    out = out * P.scale.view(1, -1, 1, 1)
                if P.shift is not None:
                    out = out + P.shift.view(1, -1, 1, 1)
                if add is not None:
                    out = out + add.float()
                out = (torch.relu(out) if P.relu else out).to(raw.dtype)
            self.data = out
        return self

    def _head_record(self):
        """(launch kwargs, bias, state) when this packed tensor is the not-yet-launched output stage of the network (deferred
        bc_head1x1 conv whose recorded epilogue is at most a per-channel shift), else None."""
        P = self._pending
        if P is None or P.conv is None or not P.conv[2].get("head") or P.scale is not None or P.add is not None or P.relu or P.interp is not None:
            return None
        P.check_source()
        return P.conv[1], P.shift, P.conv[2]

    def _sibling(self, pending) -> "TensorWrapper":
        """New TensorWrapper over the same stored data with its own pending record."""
        with _NoDispatch():
            out = self.as_subclass(torch.Tensor).as_subclass(TensorWrapper)
        out._init_metadata(self)
        out._pending = pending
        return out

    @staticmethod
    def _wrap_like(t: torch.Tensor, like: "TensorWrapper", is_blocks: bool) -> "TensorWrapper":
        with _NoDispatch():
            out = t.as_subclass(TensorWrapper)
        out._init_metadata(like)
        out._is_blocks = is_blocks
        return out

    # ------------------------------------------------------------------ dense -> packed
    def to_blocks(self, grid: torch.Tensor, grid_host: torch.Tensor = None) -> "TensorWrapper":
        """Pack the tiles selected by ``grid`` (bool, (N,1,GH,GW)).  ``grid_host`` is an optional CPU mirror of the
        grid supplied by the policy; with it the frame needs no device->host synchronisation."""
        assert not self.is_blocks
        with _NoDispatch():
            self._features._process_grid(grid, self._features_prev, grid_host)
            assert grid.dim() == 4 and self.dim() == 4
            assert self.shape[2] % grid.shape[2] == 0 and self.shape[3] % grid.shape[3] == 0
            block_size = self.shape[2] // grid.shape[2]
        return self._split(block_size)

    @property
    def fuses_dense_ops(self) -> bool:
        """True when elementwise ops on this (dense or packed) tensor are recorded and fused instead of launched."""
        return bool(fusion.ENABLED and self.is_init and self._features is not None and self._features.engine == "fused")

    def to_blocks_like(self, other: "TensorWrapper") -> "TensorWrapper":
        """Pack with the same grid as ``other``."""
        if self._pending is not None:
            self._materialize()
        self._init_metadata(other)
        self._is_blocks = False
        with _NoDispatch():
            block_size = self.shape[2] // self.get_grid().shape[2]
        return self._split(block_size)

    def _split(self, block_size: int) -> "TensorWrapper":
        assert self.is_init, "need to call process_temporal_features before splitting in blocks!"
        with timings.env("tensorwrapper/split", 10), _NoDispatch():
            if self.is_blocks:
                raise AttributeError("TensorWrapper: already split in blocks! Cannot split again.")
            if self.dim() != 4:
                raise AttributeError("TensorWrapper only supports 4D NCHW tensors!")
            N, C, H, W = self.shape
            if H % block_size != 0 or W % block_size != 0:
                raise AttributeError(f"TensorWrapper: Shape ({self.shape}) not divisibile by given block size ({block_size})!")
            grid_idx = self.get_grid_idx()
            mapping_exec = self.get_mapping_exec()
            block_size = W // grid_idx.shape[3]
            n_exec = mapping_exec.numel()
            dense = dense_layout(self.as_subclass(torch.Tensor))
            out = empty_like_layout((n_exec, C, block_size, block_size), dense)   # packed tiles keep the dense map's layout
            dyn = self._features.dyn if self._features is not None else None
            out = SplitFunction.apply(out, dense, mapping_exec, grid_idx) if dyn is None else SplitFunction.apply(out, dense, mapping_exec, grid_idx, dyn)
            return self._wrap_like(out, self, True)

    @staticmethod
    def _deferred_split(dense_map: torch.Tensor, block_size: int, feats: "BlockFeatures") -> "TensorWrapper":
        """The packed tiles of ``dense_map`` at the executed grid positions as a DEFERRED gather: the map already holds them in place
        (graph body: bc_tile_copy_indirect has just written this frame's tiles into the frame-state map), and the usual first consumer
        -- the fused stem conv -- reads its windows from the map, so the packed tensor is only materialised (bc_split from the map) if
        something else asks for its value.  The placeholder is never read or written otherwise."""
        N, C, H, W = dense_map.shape
        n_exec = feats._mapping_exec.numel()
        placeholder = empty_like_layout((n_exec, C, block_size, block_size), dense_map)
        be = get_backend()

        dkw = {} if feats.dyn is None else {"dyn": feats.dyn}

        def launch(epilogue=None, **kw):
            out = be.split(empty_like_layout((n_exec, C, block_size, block_size), dense_map), dense_map, feats._mapping_exec, feats._grid_idx, **dkw)
            return out if epilogue is None else be.affine_act(out, *epilogue, **dkw)

        with _NoDispatch():
            blocks = placeholder.as_subclass(TensorWrapper)
        blocks._init_metadata()
        blocks._features = feats
        blocks._is_blocks = True
        blocks._pending = fusion.Pending().defer_conv(launch, {}, dense_map)
        blocks._pending.conv[2]["split"] = True
        blocks._dense_map = dense_map
        return blocks

    def _is_plain_split(self) -> bool:
        """True for a packed tensor that is nothing but the not-yet-launched gather of its ``_dense_map`` (see _deferred_split)."""
        P = self._pending
        return (P is not None and P.conv is not None and bool(P.conv[2].get("split")) and P.scale is None and P.shift is None and P.add is None
                and not P.relu and P.interp is None and getattr(self, "_dense_map", None) is not None)

    # ------------------------------------------------------------------ packed -> dense
    def to_tensor(self) -> torch.Tensor:
        """Plain torch.Tensor view; packed tensors are combined (out of place) first."""
        out = self.combine() if self.is_blocks else self
        plain = out._plain()
        if (fusion.PRED_KERNEL and self.fuses_dense_ops and plain.dim() == 4 and plain.shape[1] % 32 == 0 and plain.shape[1] >= 32
                and is_nhwc(plain) and hasattr(get_backend(), "pred3x3")):
            with _NoDispatch():
                return plain.as_subclass(DenseMap)       # (see DenseMap: a plain tensor whose small-Cout 3x3 convs skip the conv library)
        return plain

    def combine_(self) -> "TensorWrapper":
        """In-place ``combine``: scatters into (and returns) the previous frame's dense map of this call site."""
        return self.combine(inplace=True)

    def combine(self, inplace: bool = False) -> "TensorWrapper":
        """Packed -> dense.  Skipped tiles carry the previous frame's values (reference: :391-443)."""
        with timings.env("tensorwrapper/combine", 4), _NoDispatch():
            if not self.is_blocks:
                raise AttributeError("TensorWrapper: Not split in blocks!")
            grid_idx = self.get_grid_idx()
            mapping_exec = self.get_mapping_exec()
            _, C, BS, _ = self.shape
            N, _, GH, GW = grid_idx.shape
            out_shape = (N, C, GH * BS, GW * BS)
            head = None if inplace else self._head_record()
            if head is not None and self._features.persistent is None and self._features.engine == "fused":
                # the network's output stage: prologue + pointwise conv + bias + out-of-place combine in ONE launch (csrc/head1x1.inc)
                kw, bias, state = head
                raw = self._raw()
                if self._features_prev:
                    prev = self._features_prev.get_features_full()
                    assert out_shape == tuple(prev.shape), (out_shape, prev.shape)
                else:
                    prev = None
                    assert mapping_exec.numel() == grid_idx.numel()
                out = torch.empty(out_shape, dtype=raw.dtype, device=raw.device, memory_format=torch.channels_last)
                get_backend().head1x1_scatter(kw["data"], kw["wpk"], kw["cout"], kw["prologue"], bias, grid_idx, mapping_exec, prev=prev, out=out)
                state["launched"] = True
                self._features.store_features_full(out)
                return self._wrap_like(out, self, False)
            if self._pending is not None:
                self._materialize()
            blocks = dense_layout(self.as_subclass(torch.Tensor))

            ps = self._features.persistent
            if ps is not None:
                # graph-capturable body: scatter into the call site's persistent map (fixed address) and hand that map
                # out, for in-place and out-of-place combines alike: a snapshot would live in the graph's private pool,
                # which is recycled at the next replay, so it could not outlive the frame any more than the map does
                # (each call site owns its map; nothing else writes it until the next frame).  Only the executed
                # tiles move: 2*n_exec*C*bs^2*E bytes instead of the 2*N*C*H*W*E of a full copy.
                buf = ps.next_map(out_shape, blocks.dtype, blocks.device, is_nhwc(blocks))
                dyn = self._dyn()
                out = CombineFunction.apply(blocks, buf, grid_idx, mapping_exec) if dyn is None else CombineFunction.apply(blocks, buf, grid_idx, mapping_exec, dyn)
                self._dense_map = out     # the map now holds these packed tiles in place: a halo window of it IS their padded form
                return self._wrap_like(out, self, False)

            if self._features_prev:
                prev = self._features_prev.get_features_full()
                assert out_shape == tuple(prev.shape), (out_shape, prev.shape)
                if is_nhwc(prev) != is_nhwc(blocks) and C > 1:
                    raise AssertionError("the memory layout of a combined map changed between frames")
                if inplace:
                    out = CombineFunction.apply(blocks, prev, grid_idx, mapping_exec)
                elif self._features.engine == "fused":
                    out = empty_like_layout(out_shape, blocks)
                    out = CombineCopyFunction.apply(blocks, prev, out, grid_idx)
                else:
                    out = CombineFunction.apply(blocks, prev.clone(), grid_idx, mapping_exec)
            else:
                # first frame of the clip: every tile is executed
                assert mapping_exec.numel() == grid_idx.numel()
                out = empty_like_layout(out_shape, blocks)
                out = CombineFunction.apply(blocks, out, grid_idx, mapping_exec)

            self._features.store_features_full(out)
            if inplace or not self._features_prev:
                self._dense_map = out     # (see above; an out-of-place combine's map may be handed to the caller and changed)
            return self._wrap_like(out, self, False)

    # ------------------------------------------------------------------ reference engine: transfer from previous frame
    def _transfer_from_prev(self):
        with timings.env("tensorwrapper/transfer", 10):
            if self._features_prev is None:
                return None
            (prev_computed, prev_padding), (prev_transfer, _) = self._features_prev.get_features()
            assert prev_transfer.shape[1:] == prev_computed.shape[1:]
            _, C, H, W = prev_transfer.shape
            transfer_idx = self._features._transfer_idx
            out = torch.empty((transfer_idx.numel(), C, H, W), dtype=prev_transfer.dtype, device=prev_computed.device)
            return TransferFunction.apply(out, prev_computed, prev_transfer, self._features_prev._grid_idx, transfer_idx, prev_padding)

    # ------------------------------------------------------------------ op routing
    @classmethod
    def __torch_function__(cls, func: Callable, types: Tuple, args: Tuple = (), kwargs: Optional[Dict] = None) -> Any:
        """Route every torch op: attribute reads and ordinary ops pass through on the packed batch; padded ops get
        their halo from neighbours / the previous frame; batch-statistics ops are re-shaped; dense-only ops raise."""
        if kwargs is None:
            kwargs = {}
        op = getattr(func, "__name__", "")
        with _NoDispatch():
            if op == "__get__":
                ret = func(*args, **kwargs)
                if not isinstance(ret, torch.Tensor):
                    return ret   # shape / dtype / device / ... : nothing to wrap
                self = _find_wrapper(args)
                if args and isinstance(args[0], TensorWrapper) and args[0]._pending is not None:
                    # tensor-valued attribute (.data, .T, .mT, .real, ...) of a lazily fused tensor: it must show the VALUE,
                    # not the stored pre-activation data / the unwritten placeholder of a deferred resampling
                    args[0]._materialize()
                    ret = func(*args, **kwargs)
                return cls._wrap_result(ret, self)

            self = _find_wrapper(args)
            if self is None:
                self = _find_wrapper(tuple(kwargs.values()))
            assert self is not None and self.is_init, "TensorWrapper used before process_temporal_features/to_blocks"

            pend = None
            if op in _FUSABLE and fusion.ENABLED and self._features is not None and self._features.engine == "fused":
                ret, pend, done = self._func_fused(op, func, args, kwargs)
                if done:
                    if isinstance(ret, TensorWrapper):
                        return ret
                    out = cls._wrap_result(ret, self)
                    out._pending = pend
                    return out
            if op not in OPS["PADDED"]:
                _materialize_args(args)
                if kwargs:
                    _materialize_args(tuple(kwargs.values()))

            if self._is_blocks and op in OPS_SPECIAL:
                if op in OPS["PADDED"]:
                    ret, pend = self._func_replace_padding(op, func, args, kwargs)
                elif op in OPS["INTERPOLATE"]:
                    ret, pend = self._func_interpolate(func, args, kwargs)
                elif op in OPS["BATCHED"]:
                    ret, pend = self._func_batched(op, func, args, kwargs)
                elif op in OPS["CHANNELONLY"]:
                    if kwargs.get("dim", None) != 1:
                        print(f"Operation {op} might behave differently with TensorWrapper when dim != 1!")
                    ret = func(*args, **kwargs)
                elif op in OPS["INCOMPATIBLE"]:
                    raise AttributeError(f"Operation {op} not supported for TensorWrapper!")
                else:
                    warnings.warn(f"Operation {op} might behave differently with TensorWrapper!")
                    ret = func(*args, **kwargs)
            else:
                ret = None
                if (op == "conv2d" and fusion.ENABLED and fusion.POINTWISE and self._features is not None and self._features.engine == "fused"
                        and isinstance(args[0], TensorWrapper)):
                    got = self._dense_pointwise_conv(args, kwargs)     # BN -> ReLU -> 1x1 conv blocks of dense (noblocks) modules
                    if got is not None:
                        ret, pend = got
                if (ret is None and op == "adaptive_avg_pool2d" and fusion.ENABLED and fusion.ADAPTIVE_POOL and self._features is not None
                        and self._features.engine == "fused" and isinstance(args[0], TensorWrapper) and not args[0]._is_blocks):
                    # pyramid pooling on a dense map (blockcopy_noblocks): one workgroup per output bin instead of the stock kernel
                    be = get_backend()
                    size = kwargs.get("output_size", args[1] if len(args) > 1 else None)
                    size = (size, size) if isinstance(size, int) else size
                    raw = args[0]._raw()
                    if (hasattr(be, "adaptive_avg_pool") and size is not None and len(size) == 2 and all(isinstance(v, int) and v > 0 for v in size)
                            and be.adaptive_avg_pool_supported(raw)):
                        ret = be.adaptive_avg_pool(raw, size)
                if ret is None:
                    if op in OPS["PADDED"]:
                        _materialize_args(args)
                    ret = func(*args, **kwargs)
            out = cls._wrap_result(ret, self)
            if pend is not None:
                out._pending = pend
            return out

    # ------------------------------------------------------------------ lazily fused elementwise ops
    def _func_fused(self, op, func, args, kwargs):
        """relu / eval batch-norm / residual add on packed tensors: record instead of launching.
        Returns (result, pending, handled)."""
        be = get_backend()
        x = args[0]
        if not isinstance(x, TensorWrapper) or x.dim() != 4 or x.dtype not in getattr(be, "supports_fusion_dtypes", ()):
            return None, None, False
        if op in ("relu", "relu_"):
            inplace = op == "relu_" or bool(kwargs.get("inplace", args[1] if len(args) > 1 else False))
            if inplace:
                if x._pending is None:
                    x._pending = fusion.Pending(relu=True)
                else:
                    x._pending.relu = True
                return x, None, True
            P = x._pending.copy() if x._pending is not None else fusion.Pending()
            P.relu = True
            return x._sibling(P), None, True
        if op == "batch_norm":
            # F.batch_norm(input, running_mean, running_var, weight=None, bias=None, training=False, momentum, eps)
            rm = kwargs.get("running_mean", args[1] if len(args) > 1 else None)
            rv = kwargs.get("running_var", args[2] if len(args) > 2 else None)
            w = kwargs.get("weight", args[3] if len(args) > 3 else None)
            b = kwargs.get("bias", args[4] if len(args) > 4 else None)
            training = kwargs.get("training", args[5] if len(args) > 5 else False)
            eps = kwargs.get("eps", args[7] if len(args) > 7 else 1e-5)
            if training or rm is None or rv is None:
                return None, None, False
            scale, shift = fusion.batchnorm_affine(rm, rv, w, b, eps)
            if x._pending is not None and not x._pending.affine_only:
                x._materialize()
            interp = conv = None
            if x._pending is not None:
                scale, shift = fusion.compose_affine(x._pending.scale, x._pending.shift, scale, shift)
                interp, conv = x._pending.interp, x._pending.conv    # a deferred producer stays the base of the record (not the placeholder)
            return x._sibling(fusion.Pending(scale=scale, shift=shift, interp=interp, conv=conv,
                                             src_guard=x._pending.src_guard if x._pending is not None else None)), None, True
        # residual add:  x (+)= y
        y = args[1] if len(args) > 1 else kwargs.get("other", None)
        alpha = kwargs.get("alpha", args[2] if len(args) > 2 else 1)
        inplace = op in ("add_", "__iadd__")
        if (alpha != 1 or not isinstance(y, torch.Tensor) or y.shape != x.shape or y.dtype != x.dtype
                or x._pending is None or not x._pending.affine_only):
            return None, None, False
        P = x._pending if inplace else x._pending.copy()
        if isinstance(y, TensorWrapper) and y._pending is not None:
            q = y._pending
            if (fusion.UPSAMPLE_EPILOGUE and P.interp is not None and P.conv is None and P.scale is None and P.shift is None
                    and q.conv is not None and q.interp is None and q.up is None and q.affine_only and not q.conv[2]["launched"]
                    and hasattr(be, "conv1x1_upsample_supported") and q.conv[0] == getattr(be, "conv1x1", None)
                    and be.conv1x1_upsample_supported(q.conv[1], P.interp)):
                # upsample(x) + conv1x1(skip), neither computed yet: the sum becomes the conv launch with the resampling in its epilogue
                R = q.copy()
                R.up, R.up_guard = P.interp, P.src_guard
                if inplace:
                    x._pending = R
                    return x, None, True
                return x._sibling(R), None, True
            if q.scale is None and q.affine_only and not q.deferred:      # (raw_y + shift_y): fold the shift, add the raw tensor
                P.shift = fusion.add_shifts(P.shift, q.shift)
                P.set_add(y._raw())
            else:
                P.set_add(y._plain())
        else:
            P.set_add(y._raw() if isinstance(y, TensorWrapper) else y)
        return (x if inplace else x._sibling(P)), None, True

    @classmethod
    def _wrap_result(cls, ret, like):
        if isinstance(ret, torch.Tensor):
            if not isinstance(ret, cls):
                ret = ret.as_subclass(cls)
            if like is not None and not ret.is_init:
                ret._init_metadata(like)
            return ret
        if isinstance(ret, (tuple, list)):
            return type(ret)(cls._wrap_result(r, like) for r in ret)
        return ret

    def _func_replace_padding(self, op, func, args, kwargs):
        """Run a padded op on the packed batch with its zero padding replaced by a gathered halo.
        Returns (result, pending-for-the-result)."""
        fuse = fusion.ENABLED and self._features.engine == "fused" and not BLOCKPAD_WITH_ZEROES
        args = list(args)
        x = args[0]
        pend_out = None
        if fuse and op == "conv2d":
            # conv bias -> pending per-channel shift on the result (folded into whatever consumes it)
            bias = kwargs["bias"] if "bias" in kwargs else (args[2] if len(args) > 2 else None)
            be = get_backend()
            if bias is not None and isinstance(x, TensorWrapper) and x.dtype in getattr(be, "supports_fusion_dtypes", ()):
                pend_out = fusion.Pending(shift=fusion.channel_vector(bias))
                if "bias" in kwargs:
                    kwargs = dict(kwargs, bias=None)
                else:
                    args[2] = None
        if BLOCKPAD_WITH_ZEROES:
            _materialize_args(args)
            return func(*args, **kwargs), pend_out
        pos = _PADDING_POS.get(op, None)
        if "padding" in kwargs:
            padding = kwargs["padding"]
        elif pos is not None and len(args) > pos:
            padding = args[pos]
        else:
            padding = 0
        zeros = 0
        if isinstance(padding, str):
            if padding != "valid":
                raise NotImplementedError(f"Only numeric paddings are supported, got {padding!r}")
            padding = 0
        elif isinstance(padding, (tuple, list)):
            zeros = (0, 0)
            if len(padding) == 1:
                padding = (padding[0], padding[0])
            if padding[0] != padding[1]:
                raise NotImplementedError(f"Only support equal paddings, got {padding}")
            padding = padding[0]
        padding = int(padding)
        if padding <= 0:
            if fuse and op == "conv2d":
                got = self._pointwise_conv(x, args, kwargs, pend_out)
                if got is not None:
                    return got
            _materialize_args(args)
            with timings.env("tensorwrapper/pad_func0", 11):
                return func(*args, **kwargs), pend_out

        feats = self._features
        if fuse and op == "conv2d" and padding == 3 and fusion.STEM_KERNEL and isinstance(x, TensorWrapper) and (x._pending is None or x._is_plain_split()):
            # network input: the padded tiles are windows of the frame-state map this packed tensor was just scattered into
            # (combine_), so window gather + 7x7 stem conv run as ONE kernel -- no halo gather, no ring cache, no layout copy
            be = get_backend()
            dm = getattr(x, "_dense_map", None)
            weight = args[1] if len(args) > 1 else kwargs.get("weight")
            cv = {k: kwargs.get(k, args[i] if len(args) > i else d) for k, i, d in (("stride", 3, 1), ("dilation", 5, 1), ("groups", 6, 1))}
            if (dm is not None and hasattr(be, "stem7x7") and isinstance(weight, torch.Tensor) and is_nhwc(weight) and feats.engine == "fused"
                    and be.stem7x7_supported(dm, weight, x.shape[2], cv["stride"], padding, cv["dilation"], cv["groups"])):
                wpk = fusion.packed_conv3x3_weight(weight, be.pack_stem7x7_weights)
                placeholder = torch.empty((x.shape[0], 64, x.shape[2] // 2, x.shape[3] // 2), dtype=dm.dtype, device=dm.device,
                                          memory_format=torch.channels_last)
                P = pend_out if pend_out is not None else fusion.Pending()
                P.defer_conv(be.stem7x7, dict(frame_state=dm, wpk=wpk, mapping_exec=feats._mapping_exec, bs=x.shape[2], **x._dyn_kw()), dm)
                return placeholder, P
        prologue = None
        residual = None      # pending residual add folded into the halo gather (with the activated tiles as a by-product)
        if isinstance(x, TensorWrapper) and x._pending is not None:
            P = x._pending
            if fuse and P.add is None and not P.deferred:
                prologue = (P.scale, P.shift, P.relu)     # folded into the halo gather; x itself stays pending
            elif fuse and feats.engine == "fused" and self._residual_gather_ok(op, x, P, args, kwargs, padding):
                residual = P
            else:
                x._materialize()
        _materialize_args(args[1:])
        data = dense_layout(x._raw() if isinstance(x, TensorWrapper) else x)
        if residual is not None:
            # end of a residual block: v = relu(raw*s + t + identity) is computed INSIDE the gather, which also emits the
            # plain v that the next shortcut needs -- one launch instead of affine pass + gather
            with timings.env("tensorwrapper/pad_residual", 10):
                padded, act = get_backend().pad_ring_add(data, dense_layout(residual.checked_add()), feats.next_ring(data, padding),
                                                         feats._grid_idx, feats._mapping_exec, padding,
                                                         (residual.scale, residual.shift, residual.relu), **x._dyn_kw())
            x._pending = None
            x.data = act
            feats._pad_memo = None
            args[0] = padded
            if "padding" in kwargs:
                kwargs = dict(kwargs, padding=zeros)
            else:
                args[pos] = zeros
            with timings.env("tensorwrapper/pad_func", 11):
                return func(*args, **kwargs), pend_out
        if feats.engine != "fused" and is_nhwc(data):
            data = data.contiguous()    # the reference decomposition is NCHW only
        grid_idx, mapping_exec = feats._grid_idx, feats._mapping_exec
        if feats.engine == "fused" and fuse and op == "conv2d" and padding in (1, 2):
            # 3x3 convs (stride 1 / 2, or dilation 2 with padding 2): ONE hand-written MFMA kernel gathers the halo and convolves (no padded tensor, no
            # library conv) wherever it is the faster route for this layer shape; else halo gather + library conv
            be = get_backend()
            plan = self._conv3x3_plan(be, data, args, kwargs, grid_idx, mapping_exec, func, padding)
            if plan is not None:
                weight = args[1] if len(args) > 1 else kwargs["weight"]
                wpk = fusion.packed_conv3x3_weight(weight, be.pack_conv3x3_weights)   # cached per parameter object
                ring = feats.next_ring(data, padding)
                feats._pad_memo = None
                stride = self._conv_stride(args, kwargs)
                dil = {} if padding == 1 else {"dilation": padding}        # (padding 2 reaches here only as padding = dilation = 2)
                if fusion.DEFER_CONV and data.dtype in getattr(be, "supports_fusion_dtypes", ()):
                    # deferred: the launch happens when the value is needed, carrying the elementwise work recorded by then (bias,
                    # residual add, ReLU) as its epilogue.  The placeholder is never read or written; `data` is held by the record.
                    placeholder = empty_like_layout((data.shape[0], weight.shape[0], data.shape[2] // stride, data.shape[3] // stride), data)
                    P = pend_out if pend_out is not None else fusion.Pending()
                    P.defer_conv(be.conv3x3_ring, dict(data_exec=data, ring=ring, wpk=wpk, cout=weight.shape[0], grid_idx=grid_idx,
                                                       mapping_exec=mapping_exec, prologue=prologue, cfg=plan, stride=stride, **dil, **x._dyn_kw()), data,
                                 registry=feats._deferred)
                    return placeholder, P
                with timings.env("tensorwrapper/conv3x3_fused", 10):
                    return be.conv3x3_ring(data, ring, wpk, weight.shape[0], grid_idx, mapping_exec, prologue, None, cfg=plan,
                                           stride=stride, **dil, **x._dyn_kw()), pend_out
        if feats.engine == "fused" and fuse and op == "max_pool2d" and padding == 1:
            # the ResNet stem pool (3x3, stride 2): halo gather + max in one kernel, no padded tensor
            be = get_backend()
            mp = {k: kwargs.get(k, args[i] if len(args) > i else d) for k, i, d in
                  (("kernel_size", 1, None), ("stride", 2, None), ("dilation", 4, 1), ("ceil_mode", 5, False), ("return_indices", 6, False))}
            one = lambda v: v if isinstance(v, int) else (v[0] if isinstance(v, (tuple, list)) and len(set(v)) == 1 else None)
            if (hasattr(be, "maxpool3x3s2_ring") and one(mp["kernel_size"]) == 3 and one(mp["stride"]) == 2 and one(mp["dilation"]) == 1
                    and not mp["ceil_mode"] and not mp["return_indices"] and be.maxpool3x3s2_supported(data)):
                ring = feats.next_ring(data, padding)
                feats._pad_memo = None
                with timings.env("tensorwrapper/maxpool_fused", 10):
                    return be.maxpool3x3s2_ring(data, ring, grid_idx, mapping_exec, prologue, **x._dyn_kw()), pend_out
        if feats.engine == "fused":
            # consecutive padded ops on the same tensor (e.g. the three CSP head branches on the 768-channel map)
            # share ONE halo gather and ring cache; the memo holds the source, so its address cannot be recycled
            # (the memo keeps the prologue tensors themselves and compares identities: an id() alone could be recycled)
            pro_key = None if prologue is None else (prologue[0], prologue[1], bool(prologue[2]))
            m = feats._pad_memo
            same_pro = m is not None and ((m[2] is None and pro_key is None) or (
                m[2] is not None and pro_key is not None and m[2][0] is pro_key[0] and m[2][1] is pro_key[1] and m[2][2] == pro_key[2]))
            if (m is not None and m[0].data_ptr() == data.data_ptr() and m[0].shape == data.shape and m[0].stride() == data.stride()
                    and m[0]._version == data._version and m[1] == padding and same_pro):
                args[0] = m[3]
            else:
                ring = feats.next_ring(data, padding)
                with timings.env("tensorwrapper/pad", 10):
                    args[0] = pad_ring(data, ring, grid_idx, mapping_exec, padding, prologue, feats.dyn)
                feats._pad_memo = (data, padding, pro_key, args[0])
        else:
            data_transfer = self._transfer_from_prev()
            if data_transfer is None:
                _, C, H, W = data.shape
                data_transfer = torch.empty((0, C, H, W), dtype=data.dtype, device=data.device)
            feats.store_features(data, data_transfer, padding)
            with timings.env("tensorwrapper/pad", 10):
                args[0] = pad(data, data_transfer, grid_idx, mapping_exec, padding)

        if "padding" in kwargs:
            kwargs = dict(kwargs, padding=zeros)
        else:
            args[pos] = zeros
        with timings.env("tensorwrapper/pad_func", 11):
            return func(*args, **kwargs), pend_out

    @staticmethod
    def _conv_stride(args, kwargs) -> int:
        st = kwargs.get("stride", args[3] if len(args) > 3 else 1)
        return st if isinstance(st, int) else st[0]

    def _conv3x3_plan(self, be, data, args, kwargs, grid_idx, mapping_exec, func, padding=1):
        """None = halo gather + library conv; int = fused halo+conv kernel with that decomposition (fusion.conv3x3_plan)."""
        weight = args[1] if len(args) > 1 else kwargs.get("weight")
        cv = {k: kwargs.get(k, args[i] if len(args) > i else d) for k, i, d in (("stride", 3, 1), ("dilation", 5, 1), ("groups", 6, 1))}
        if not (isinstance(weight, torch.Tensor) and hasattr(be, "conv3x3_ring")
                and be.conv3x3_supported(data, weight, cv["stride"], padding, cv["dilation"], cv["groups"])):
            return None
        dil = 1 if padding == 1 else 2
        n_exec, cin, bs = data.shape[0], data.shape[1], data.shape[2]
        cout, n_total = weight.shape[0], grid_idx.numel()
        n_plan = self._features.plan_n_exec or n_exec        # (dynamic graph: form chosen for the expected count, see BlockFeatures.plan_n_exec)
        stride = self._conv_stride(args, kwargs)

        def tuner():
            if not hasattr(be, "conv3x3_candidates") or not data.is_cuda:
                return None
            w_plain = weight.as_subclass(torch.Tensor) if isinstance(weight, TensorWrapper) else weight
            wpk = fusion.packed_conv3x3_weight(weight, be.pack_conv3x3_weights)
            scratch = torch.zeros((n_total, cin, 4 * dil * bs), dtype=data.dtype, device=data.device)   # a ring nobody else reads
            dkw = {} if dil == 1 else {"dilation": dil}
            lib0 = lambda: torch.nn.functional.conv2d(be.pad_ring(data, scratch, grid_idx, mapping_exec, dil, None), w_plain.detach(), stride=stride, dilation=dil)
            lib = lib0
            if fusion.TUNE_EPILOGUE_COST:      # (see fusion.TUNE_EPILOGUE_COST: the elementwise pass the library route needs after the conv)
                zero = torch.zeros(cout, dtype=torch.float32, device=data.device)
                lib = lambda: be.affine_act(lib0(), None, zero, None, True)
            routes = {"library": lib}
            for c in be.conv3x3_candidates(n_exec, cin, cout, bs, data.element_size(), stride, **dkw):
                routes[str(c)] = (lambda c_: lambda: be.conv3x3_ring(data, scratch, wpk, cout, grid_idx, mapping_exec, None, None, cfg=c_, stride=stride, **dkw))(c)
            return be.time_routes(routes)

        def candidates():
            if not hasattr(be, "conv3x3_candidates"):
                return []
            return be.conv3x3_candidates(n_exec, cin, cout, bs, data.element_size(), stride, **({} if dil == 1 else {"dilation": dil}))

        # (plan-table key: kernel-size field 3 for the plain form, 13 for dilation 2)
        plan = fusion.conv3x3_plan(n_plan, bs, cin, cout, n_total, data.dtype, tuner, stride, ks=3 if dil == 1 else 13, candidates=candidates)
        if plan is None and self._features.dyn is not None:
            plan = -1       # (a library conv on a ceiling-sized packed tensor would compute every row: the own kernel reads the count)
        return plan

    def _dense_pointwise_conv(self, args, kwargs):
        """conv2d on a dense (non-packed) TensorWrapper: 1x1 convs take the fused one-tap kernel; returns (result, pending) or None."""
        args = list(args)
        if "padding" in kwargs:
            padding = kwargs["padding"]
        else:
            padding = args[4] if len(args) > 4 else 0
        if isinstance(padding, (tuple, list)):
            padding = padding[0] if len(set(padding)) == 1 else -1
        if padding != 0:
            return None
        pend_out = None
        bias = kwargs["bias"] if "bias" in kwargs else (args[2] if len(args) > 2 else None)
        got = self._pointwise_conv(args[0], args, kwargs, None, bias=bias)
        return got

    def _pointwise_conv(self, x, args, kwargs, pend_out, bias=None):
        """1x1 / pad 0 conv through bc_conv1x1_nhwc where that is the faster route: the pending BN / ReLU of ``x`` becomes the
        kernel's prologue, the launch is deferred so that what is recorded after it becomes its epilogue.  None = library route."""
        be = get_backend()
        weight = args[1] if len(args) > 1 else kwargs.get("weight")
        if (not fusion.POINTWISE or not hasattr(be, "conv1x1") or not isinstance(weight, torch.Tensor) or not isinstance(x, TensorWrapper)
                or x.dim() != 4 or tuple(weight.shape[2:]) != (1, 1)):
            return None
        cv = {k: kwargs.get(k, args[i] if len(args) > i else d) for k, i, d in (("stride", 3, 1), ("dilation", 5, 1), ("groups", 6, 1))}
        stride = self._conv_stride(args, kwargs)
        raw = x._raw()
        if (fusion.HEAD_KERNEL and fusion.DEFER_CONV and x._is_blocks and hasattr(be, "head1x1") and weight.shape[0] <= 32
                and be.head1x1_supported(raw, weight, cv["stride"], 0, cv["dilation"], cv["groups"])
                and raw.dtype in getattr(be, "supports_fusion_dtypes", ())):
            # the network's output stage (few output channels): deferred like every fused conv; if the value is consumed by an
            # out-of-place combine -- the normal end of a frame -- prologue, conv, bias and scatter+copy are ONE launch (combine())
            P = x._pending
            prologue = None
            if P is not None and P.add is None and not P.deferred:
                prologue = (P.scale, P.shift, P.relu)
            elif P is not None:
                x._materialize()
            if pend_out is None and bias is not None:
                pend_out = fusion.Pending(shift=fusion.channel_vector(bias))
            data = dense_layout(x._raw())
            wpk = fusion.packed_conv3x3_weight(weight, be.pack_head1x1_weights)
            cout = weight.shape[0]
            placeholder = torch.empty((data.shape[0], cout, data.shape[2], data.shape[3]), dtype=data.dtype, device=data.device,
                                      memory_format=torch.channels_last)
            Pn = pend_out if pend_out is not None else fusion.Pending()
            Pn.defer_conv(be.head1x1, dict(data=data, wpk=wpk, cout=cout, prologue=prologue, **x._dyn_kw()), data)
            Pn.conv[2]["head"] = True
            return placeholder, Pn
        if not be.conv1x1_supported(raw, weight, cv["stride"], 0, cv["dilation"], cv["groups"]):
            return None
        P = x._pending
        foldable = P is not None and P.add is None and not P.deferred
        n_px, cin, cout = raw.shape[0] * raw.shape[2] * raw.shape[3], raw.shape[1], weight.shape[0]
        dyn_rows = x._dyn() is not None
        n_px_plan = (x._features.plan_n_exec or raw.shape[0]) * raw.shape[2] * raw.shape[3] if dyn_rows else n_px

        def tuner():
            if not raw.is_cuda:
                return None
            w_plain = weight.as_subclass(torch.Tensor) if isinstance(weight, TensorWrapper) else weight
            wpk = fusion.packed_conv3x3_weight(weight, be.pack_conv3x3_weights)
            src = dense_layout(raw)
            pro = (P.scale, P.shift, P.relu) if foldable else None
            lib0 = (lambda: torch.nn.functional.conv2d(be.affine_act(src, pro[0], pro[1], None, pro[2]), w_plain.detach(), stride=stride)) if pro is not None \
                else (lambda: torch.nn.functional.conv2d(src, w_plain.detach(), stride=stride))
            lib = lib0
            if fusion.TUNE_EPILOGUE_COST:
                # what follows a conv in a CNN (bias / folded BN, residual add, ReLU) rides in the fused kernel's epilogue but costs
                # the library route one more elementwise pass over the result: price that pass in
                zero = torch.zeros(cout, dtype=torch.float32, device=src.device)
                lib = lambda: be.affine_act(lib0(), None, zero, None, True)
            routes = {"library": lib}
            for c in be.conv1x1_candidates(src, cout, stride):
                routes[str(c)] = (lambda c_: lambda: be.conv1x1(src, wpk, cout, pro, None, cfg=c_, stride=stride))(c)
            return be.time_routes(routes)

        plan = fusion.conv3x3_plan(max(1, n_px_plan // 64), 8, cin, cout, int(foldable), raw.dtype, tuner, stride, ks=1)
        if plan is None and dyn_rows:
            plan = -1       # (see _conv3x3_plan: the library route would run on every row of the ceiling-sized tensor)
        if plan is None:
            return None
        if stride != 1 and plan >= 0:
            # the plan key of a pointwise conv carries the pixel count, not the tile size -- enough for stride 1 (any 8x8 re-tiling), but
            # a stride-2 launch works on the REAL tiles: a decomposition measured on 16-pixel tiles need not cover 8-pixel ones with the
            # same pixel count (found by the differential fuzz: plans measured live in one geometry met another one).  Checked once per shape.
            vkey = (plan, n_px, raw.shape[2], cin, cout, raw.dtype)
            ok = _S2_PLAN_OK.get(vkey)
            if ok is None:
                ok = _S2_PLAN_OK[vkey] = (not hasattr(be, "conv1x1_candidates")) or plan in be.conv1x1_candidates(dense_layout(raw), cout, stride)
            if not ok:
                plan = -1
        if pend_out is None and bias is not None:
            if raw.dtype not in getattr(be, "supports_fusion_dtypes", ()):
                return None
            pend_out = fusion.Pending(shift=fusion.channel_vector(bias))
        prologue = None
        if foldable:
            prologue = (P.scale, P.shift, P.relu)
        elif P is not None:
            x._materialize()
        data = dense_layout(x._raw())
        wpk = fusion.packed_conv3x3_weight(weight, be.pack_conv3x3_weights)
        launch_kw = dict(data=data, wpk=wpk, cout=cout, prologue=prologue, cfg=plan, stride=stride, **x._dyn_kw())
        if fusion.DEFER_CONV and data.dtype in getattr(be, "supports_fusion_dtypes", ()):
            placeholder = torch.empty((data.shape[0], cout, data.shape[2] // stride, data.shape[3] // stride), dtype=data.dtype, device=data.device,
                                      memory_format=torch.channels_last)
            Pn = pend_out if pend_out is not None else fusion.Pending()
            Pn.defer_conv(be.conv1x1, launch_kw, data)
            return placeholder, Pn
        return be.conv1x1(epilogue=None, **launch_kw), pend_out

    def _residual_gather_ok(self, op, x, P, args, kwargs, padding) -> bool:
        """Can the pending residual add of ``x`` be folded into this padded op's halo gather?"""
        be = get_backend()
        if not hasattr(be, "pad_ring_add") or P.add is None or P.deferred or not isinstance(P.add, torch.Tensor):
            return False
        raw = x._raw()
        if not (is_nhwc(raw) and be.pad_ring_add_supported(dense_layout(raw), dense_layout(P.add))):
            return False
        if op == "conv2d" and padding in (1, 2):
            # layers that go to the fused MFMA conv keep the plain route (materialise, then conv without prologue)
            feats = self._features
            if self._conv3x3_plan(be, dense_layout(raw), args, kwargs, feats._grid_idx, feats._mapping_exec, None, padding) is not None:
                return False
        return True

    def _func_interpolate(self, func, args, kwargs):
        """Resampling runs per tile on the packed batch, i.e. WITHOUT halo: a tile's border pixels are interpolated
        from the tile alone.  (The reference reaches the same arithmetic through a trilinear re-expression on a
        (1,B,C,h,w) view, core/tensorwrapper.py:577-598, because the stock bilinear kernel loops over tiles x
        channels inside each thread.)  Bilinear goes to the library's fully parallel per-tile kernel; every other
        mode runs as the stock op on the packed batch."""
        data = args[0].as_subclass(torch.Tensor)
        mode = kwargs.get("mode", args[3] if len(args) > 3 else "nearest")
        if func.__name__ == "upsample_bilinear":
            mode, kwargs = "bilinear", dict(kwargs, align_corners=True)
        be = get_backend()
        if (mode != "bilinear" or data.dim() != 4 or kwargs.get("antialias", False)
                or data.dtype not in getattr(be, "supports_interp_dtypes", ())):
            return func(*args, **kwargs), None
        size = kwargs.get("size", args[1] if len(args) > 1 else None)
        scale = kwargs.get("scale_factor", args[2] if len(args) > 2 else None)
        align = bool(kwargs.get("align_corners", False))
        recompute = bool(kwargs.get("recompute_scale_factor", False))
        h, w = data.shape[2], data.shape[3]
        if size is not None:
            H, W = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
            sh = sw = None
        else:
            sh, sw = (float(scale), float(scale)) if not isinstance(scale, (tuple, list)) else (float(scale[0]), float(scale[1]))
            H, W = int(h * sh), int(w * sw)   # floor, as torch
            if recompute:
                sh = sw = None

        def resolved(in_size, out_size, s):
            if align:
                return np.float32(in_size - 1) / np.float32(out_size - 1) if out_size > 1 else np.float32(0)
            return np.float32(1.0 / s) if (s is not None and s > 0) else np.float32(in_size) / np.float32(out_size)

        src, rh, rw = dense_layout(data), resolved(h, H, sh), resolved(w, W, sw)
        if (fusion.ENABLED and self._features.engine == "fused" and hasattr(be, "interp_epilogue_supported")
                and be.interp_epilogue_supported(src) and src.dtype in getattr(be, "supports_fusion_dtypes", ())):
            # deferred: the launch happens when the value is needed, with whatever elementwise work was recorded by then
            # as its epilogue (decoder: "upsample, += skip" -> one kernel).  The placeholder is never read or written.
            placeholder = empty_like_layout((src.shape[0], src.shape[1], H, W), src)
            return placeholder, fusion.Pending(interp=(src, H, W, align, rh, rw))
        return be.interp_bilinear(src, H, W, align, rh, rw, **args[0]._dyn_kw()), None

    def _func_batched(self, op, func, args, kwargs):
        """Ops with per-sample statistics (group_norm): fold the tile axis into the spatial axis so statistics
        run over all executed tiles of the (batch-size-1) frame, as the reference does (:600-633).
        Returns (result, pending-for-the-result)."""
        args = list(args)
        if self._features is not None and self._features.dyn is not None:
            raise NotImplementedError(f"{op} over all executed tiles needs the executed-tile count on the host (statistics run over exactly those "
                                      "tiles): not available with a device-side count (block_graph=2); use block_graph=1")
        data = args[0].as_subclass(torch.Tensor)
        B, C, H, W = data.shape
        if op == "group_norm" and fusion.ENABLED and fusion.GROUP_NORM and self._features.engine == "fused":
            # channels-last packed tiles: the op is a per-channel affine map whose coefficients cost one read of the tensor
            # (bc_group_norm_affine_nhwc); recorded as pending work, i.e. applied by whatever consumes the result
            be = get_backend()
            gn = {k: kwargs.get(k, args[i] if len(args) > i else d) for k, i, d in (("num_groups", 1, None), ("weight", 2, None), ("bias", 3, None), ("eps", 4, 1e-5))}
            if (hasattr(be, "group_norm_affine") and gn["num_groups"] is not None and data.dtype in getattr(be, "supports_fusion_dtypes", ())
                    and be.group_norm_affine_supported(dense_layout(data), gn["num_groups"])):
                vec = lambda v: None if v is None else fusion.channel_vector(v)
                scale, shift = be.group_norm_affine(dense_layout(data), gn["num_groups"], vec(gn["weight"]), vec(gn["bias"]), gn["eps"])
                return args[0]._sibling(fusion.Pending(scale=scale, shift=shift)), None     # (a sibling: the input tensor keeps its own value)
        args[0] = data.permute(1, 0, 2, 3).reshape(1, C, B * H * W, 1)
        out = func(*args, **kwargs)
        return out.reshape(C, B, H, W).permute(1, 0, 2, 3).contiguous(), None
