"""Binding of the block ops to libblockcopy_hip.so (C ABI: include/blockcopy_hip.h) through ctypes.

The product has exactly one compute backend: the gfx950 HIP library.  There is no CPU or eager-PyTorch
fallback -- if the library is missing, or a tensor is not on the GPU, the ops raise (the reference is the
same: "NVIDIA CUDA-capable GPU (no CPU support)", README.md:21; asserts in utils/cuda.py:42-48).

``set_backend`` exists so that the *test-suite* can drive the host logic (state machine, op routing,
index tables, ring-cache bookkeeping) on a machine without a GPU by injecting a checker backend built on
the CPU oracle; nothing in this package ever does that by itself.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("BLOCKCOPY_HIP_LIB", os.path.join(_PKG_ROOT, "lib", "libblockcopy_hip.so"))

OP_SPLIT, OP_COMBINE, OP_TRANSFER, OP_PAD, OP_COMBINE_COPY, OP_PAD_RING, OP_GRID_TABLES, OP_INTERP, OP_AFFINE = range(9)
OP_NAMES = ("split", "combine", "transfer", "pad", "combine_copy", "pad_ring", "grid_tables", "interp", "affine", "nms", "conv3x3", "head1x1", "pred3x3")
_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
ABI_VERSION = 2


class BlockCopyBackendError(RuntimeError):
    code = None   # the library's return code (negative BC_ERR_*, positive hipError_t) when the error came from a launch


BC_ERR_SHAPE = -2


def is_nhwc(x: torch.Tensor) -> bool:
    """True for a 4-D tensor whose memory is channels-last (and not also plain-contiguous, e.g. C == 1)."""
    return x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)


def dense_layout(x: torch.Tensor) -> torch.Tensor:
    """x itself if it is NCHW- or channels-last-contiguous, else a contiguous copy."""
    if x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        return x
    return x.contiguous()


def empty_like_layout(shape, like: torch.Tensor) -> torch.Tensor:
    fmt = torch.channels_last if is_nhwc(like) else torch.contiguous_format
    return torch.empty(shape, dtype=like.dtype, device=like.device, memory_format=fmt)


class PinnedRing:
    """A few reusable pinned host buffers of one shape for per-frame staging (index tables, grids): page-locking memory costs
    tens of microseconds per allocation, so the engine must not do it every frame.  ``next()`` hands out the buffers round
    robin; before a buffer is reused, the event recorded after its last asynchronous upload is waited for (normally long
    past), so a copy still in flight is never overwritten.  On CPU-only runs (test tier) the buffers are ordinary tensors."""

    def __init__(self, numel: int, dtype, pinned: bool, depth: int = 4):
        self.bufs = [torch.empty(numel, dtype=dtype, pin_memory=pinned) for _ in range(depth)]
        self.events = [None] * depth
        self.pos = 0
        self.pinned = pinned

    def next(self):
        i = self.pos
        self.pos = (i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
        self._last = i
        return self.bufs[i]

    def uploaded(self):
        """Call right after enqueueing the H->D copy of the buffer returned by the last ``next()``."""
        if self.pinned:
            ev = self.events[self._last] or torch.cuda.Event()
            ev.record()
            self.events[self._last] = ev


_pinned_rings = {}


def pinned_ring(numel: int, dtype, pinned: bool) -> PinnedRing:
    key = (int(numel), dtype, bool(pinned))
    r = _pinned_rings.get(key)
    if r is None:
        r = _pinned_rings[key] = PinnedRing(numel, dtype, pinned)
    return r


def _ok(x: torch.Tensor, *dtypes) -> bool:
    # same contract as the reference's cudaok(), utils/cuda.py:42-48 (plus channels-last as a second dense layout)
    assert x.is_cuda, "blockcopy ops need GPU tensors (no CPU path)"
    assert x.is_contiguous() or is_nhwc(x), "blockcopy ops need contiguous NCHW or channels-last tensors"
    assert not dtypes or x.dtype in dtypes, (x.dtype, dtypes)
    return True


def load_library(path: str = None) -> ctypes.CDLL:
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise BlockCopyBackendError(
            f"{path} not found: build it with `python blockcopy-video-processing-pytorch_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no fallback backend.")
    lib = ctypes.CDLL(path)
    i, p, u = ctypes.c_int, ctypes.c_void_p, ctypes.c_uint
    sig = {
        "bc_split": [p, p, p] + [i] * 7 + [p],
        "bc_combine": [p, p, p] + [i] * 7 + [p],
        "bc_combine_copy": [p, p, p, p] + [i] * 6 + [p],
        "bc_combine_copy_indirect": [p, p, p] + [i] * 7 + [p],
        "bc_combine_copy_cells": [p] + [i] * 7,
        "bc_tile_copy_indirect": [p, p, p, p] + [i] * 8 + [p],
        "bc_head1x1_scatter_nhwc": [p] * 7 + [i] * 8 + [p, p, i, p, i, p],
        "bc_transfer": [p, p, p, p] + [i] * 8 + [p],
        "bc_pad": [p, p, p, p, p] + [i] * 8 + [p],
        "bc_pad_ring": [p, p, p, p, p] + [i] * 8 + [p],
        "bc_grid_tables": [p, i, p, p, p, p, p, p],
        "bc_grid_tables_host": [p, i, p, p, p, p],
        "bc_interp_bilinear": [p, p, ctypes.c_longlong, i, i, i, i, i, ctypes.c_float, ctypes.c_float, i, p],
        "bc_pad_ring_act": [p, p, p, p, p] + [i] * 8 + [p, p, i, p],
        "bc_pad_ring_nhwc": [p, p, p, p, p] + [i] * 9 + [p, p, i, p],
        "bc_affine_act_nhwc": [p, p, p, p, p, i, ctypes.c_longlong, i, i, p],
        "bc_maxpool3x3s2_ring_nhwc": [p, p, p, p, p] + [i] * 7 + [p, p, i, p],
        "bc_pad_ring_add_nhwc": [p, p, p, p, p, p, p] + [i] * 8 + [p, p, i, p],
        "bc_conv3x3_ring_nhwc": [p, p, p, p, p, p] + [i] * 8 + [p, p, i, p, p, p, i, p],
        "bc_conv3x3s2_ring_nhwc": [p, p, p, p, p, p] + [i] * 8 + [p, p, i, p, p, p, i, p],
        "bc_conv3x3_candidates": [i, i, i, i, i, i, ctypes.POINTER(i), i],
        "bc_conv3x3_dil_ring_nhwc": [p, p, p, p, p, p] + [i] * 9 + [p, p, i, p, p, p, i, p],
        "bc_conv3x3_dil_candidates": [i, i, i, i, i, i, ctypes.POINTER(i), i],
        "bc_pred3x3_nhwc": [p, p, p, p, i, i, i, i, i, i, p],
        "bc_spp_levels_nhwc": [p, p, p, p, p, i, i, i, i, i, ctypes.POINTER(i), i, p],
        "bc_spp_levels_n_nhwc": [p, p, p, p, p, i, i, i, i, i, i, ctypes.POINTER(i), i, p],
        "bc_spp_fuse_nhwc": [p, p, p, p, p, p, i, i, i, i, i, ctypes.POINTER(i), i, i, p],
        "bc_spp_fuse_n_nhwc": [p, p, p, p, p, p, i, i, i, i, i, i, ctypes.POINTER(i), i, i, p],
        "bc_spp_fuse_packed_nhwc": [p, p, p, p, p, p, p, i, i, i, i, i, i, i, ctypes.POINTER(i), i, i, p],
        "bc_stem7x7s2_nhwc": [p, p, p, p] + [i] * 7 + [p, p, p, i, p],
        "bc_conv1x1_nhwc": [p, p, p] + [i] * 6 + [p, p, i, p, p, p, i, p],
        "bc_conv1x1_candidates": [i, i, i, i, i, i, ctypes.POINTER(i), i],
        "bc_conv_upsample_arm": [p, i, i, i, ctypes.c_float, ctypes.c_float],
        "bc_nms_sorted_dev": [p, i, p, ctypes.c_float, p, p, p, p],
        "bc_csp_decode": [p, p, p, p, p, i, i, i, ctypes.c_float, i, i, ctypes.c_float, p, p, p],
        "bc_csp_score_monotone": [p, p],
        "bc_csp_topk_decode": [p, i, p, p, ctypes.c_longlong, ctypes.c_longlong, i, i, i, i, ctypes.c_float, i, i, ctypes.c_float, p, p, p, p],
        "bc_interp_bilinear_nhwc": [p, p, ctypes.c_longlong, i, i, i, i, i, i, ctypes.c_float, ctypes.c_float, i, p],
        "bc_upsample_argmax": [p, p, i, i, i, i, i, i, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, i,
                               ctypes.c_float, ctypes.c_float, i, p],
        "bc_interp_bilinear_act_nhwc": [p, p, ctypes.c_longlong, i, i, i, i, i, i, ctypes.c_float, ctypes.c_float, i, p, p, p, i, p],
        "bc_affine_act": [p, p, p, p, p, i, ctypes.c_longlong, i, ctypes.c_longlong, i, p],
        "bc_bn_train_fwd": [p, p, i, i, ctypes.c_longlong, p, p, p, p, p, p, p, ctypes.c_float, ctypes.c_float, i, p, ctypes.c_longlong, p],
        "bc_bn_train_stats_nhwc": [p, ctypes.c_longlong, i, i, ctypes.c_float, p, p, p, p, p, ctypes.c_float, p, p, p, p, p, ctypes.c_longlong, p],
        "bc_adaptive_avg_pool_nhwc": [p, p, i, i, i, i, i, i, i, p],
        "bc_group_norm_affine_nhwc": [p, ctypes.c_longlong, i, i, i, ctypes.c_float, p, p, p, p, p, ctypes.c_longlong, p],
        "bc_nms_sorted": [p, i, ctypes.c_float, p, p, p, p],
        "bc_l2norm_cat_nhwc": [p, p, p, ctypes.c_longlong, i, i, i, ctypes.c_float, i, p],
        "bc_l2norm_cat_deconv_nhwc": [p, p, p, p, i, i, i, i, i, i, i, ctypes.c_float, i, p],
        "bc_policy_step": [p, i, ctypes.c_ulonglong, ctypes.c_ulonglong, i, i, p, p, p, p, p, p],
        "bc_policy_features": [p, i, i, i, p, p, p, p, p],
        "bc_pn_conv_nhwc": [p, p, p] + [i] * 10 + [p, p, i, p, p, i, p, ctypes.c_longlong, i, p],
        "bc_pn_wgrad_nhwc": [p, p, ctypes.c_longlong, p, p] + [i] * 9 + [p, p, i, p],
        "bc_pn_bn_finalize": [p, ctypes.c_longlong, i, ctypes.c_double, p, p, ctypes.c_float, ctypes.c_float, p, p, p, p, p, p, p, p],
        "bc_pn_join": [p, p, p, p, p, p, p, i, i, ctypes.c_longlong, p],
        "bc_pn_arm_bn": [p, p, p, ctypes.c_double, ctypes.c_float, i, p],
        "bc_pn_join_acc": [p, p, p, p, p, p, p, p, p, ctypes.c_double, ctypes.c_float, i, i, ctypes.c_longlong, p],
        "bc_pn_head_fwd_acc": [p, p, p, p, p, ctypes.c_double, ctypes.c_float, p, p, i, i, i, i, p],
        "bc_pn_bn_layer_bytes": [],
        "bc_pn_bn_finalize_acc": [p, i, p],
        "bc_pn_bn_bwd": [p, p, p, p, p, p, p, p, i, p, p, p, p, p, i, ctypes.c_longlong, p],
        "bc_pn_head_fwd": [p, p, p, p, p, p, i, i, i, i, p],
        "bc_pn_head_bwd": [p, p, p, p, p, p, p, p, i, i, i, i, p],
        "bc_pn_infogain": [p, p, p, i, i, i, i, i] + [ctypes.c_longlong] * 4 + [i, i, ctypes.c_float, ctypes.c_float, p],
        "bc_pn_reward_seed": [p, p, p, p, p, p, p, ctypes.c_double, ctypes.c_double, ctypes.c_double, i, i, i, i, i, p],
        "bc_pn_rmsprop": [p, p, p, p, ctypes.c_longlong] + [ctypes.c_float] * 5 + [p],
        "bc_pn_sync_params": [p, p, p, i, i, p],
        "bc_pn_seg_bytes": [],
        "bc_pn_wgrad_groups": [i, i, i, i, i],
        "bc_pn_update": [p, p, p, p, p, p, p, i] + [ctypes.c_float] * 5 + [p],
        "bc_pn_set_stamps": [p],
        "bc_pn_probs": [p, p, p, p, i, p],
        "bc_pn_features_nhwc": [p, i, i, i, i, p, p, p, p, p],
        "bc_dyn_set": [p, i],
        "bc_tune_set": [ctypes.c_char_p, i],
        "bc_tune_get": [ctypes.c_char_p, ctypes.POINTER(i)],
        "bc_tune_set_ptr": [ctypes.c_char_p, p],
        "bc_abi_version": [],
        "bc_prof_enable": [u],
        "bc_prof_reset": [],
        "bc_prof_read": [i, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
        "bc_prof_read_aux": [i, ctypes.POINTER(ctypes.c_double)],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = i
    for name, argtypes in {"bc_pn_conv_partials": [i, i, i, i], "bc_pn_wgrad_workspace": [i] * 6, "bc_pn_bn_bwd_partials": [ctypes.c_longlong]}.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_longlong
    lib.bc_error_string.argtypes = [i]
    lib.bc_error_string.restype = ctypes.c_char_p
    lib.bc_op_name.argtypes = [i]
    lib.bc_op_name.restype = ctypes.c_char_p
    if lib.bc_abi_version() != ABI_VERSION:
        raise BlockCopyBackendError(f"ABI mismatch: library {lib.bc_abi_version()} vs binding {ABI_VERSION}")
    return lib


class HipBackend:
    """The product backend: raw device pointers into libblockcopy_hip.so on torch's current stream."""

    name = "hip"

    def __init__(self, path: str = None):
        self.lib = load_library(path)
        self.lib.bc_tune_set(b"stem_split", int(os.environ.get("BLOCKCOPY_STEM_SPLIT", "1")))      # (pack_stem7x7_weights provides the split streams)
        self._conv_cfg = None          # value last handed to the library
        self._conv_cfg_pinned = -1     # value pinned through tune("conv2_cfg", ...) for calls that do not choose themselves

    # -- helpers
    def _check(self, rc: int, op: str):
        if rc != 0:
            err = BlockCopyBackendError(f"bc_{op} failed: {self.lib.bc_error_string(rc).decode()} (code {rc})")
            err.code = rc
            raise err

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def _arm(self, dyn):
        """``dyn`` = (n_exec_dev: device int32[1], ceiling) or None: make the NEXT launch read its executed-tile count from the
        device (include/blockcopy_hip.h bc_dyn_set; the launcher consumes the arming at entry, whatever it then does)."""
        if dyn is not None:
            ptr, ceiling = dyn
            assert _ok(ptr, torch.int32) and ptr.numel() >= 1
            self._check(self.lib.bc_dyn_set(ptr.data_ptr(), int(ceiling)), "dyn_set")

    # -- A. reference operator boundary ---------------------------------------------------------------
    def split(self, blocks, image, mapping_exec, grid_idx, dyn=None):
        """blocks <- executed tiles of image.  reference: SplitFunction.forward, utils/block_funcs.py:12-49."""
        assert _ok(blocks) and _ok(image, blocks.dtype) and _ok(mapping_exec, torch.int32) and _ok(grid_idx, torch.int32)
        assert blocks.dim() == 4 and image.dim() == 4 and blocks.shape[2] == blocks.shape[3]
        n_exec, C, bs, _ = blocks.shape
        N, C_img, H, W = image.shape
        _, _, GH, GW = grid_idx.shape
        assert C == C_img and GH * bs == H and GW * bs == W, (blocks.shape, image.shape, grid_idx.shape)
        assert n_exec == mapping_exec.numel()
        # (1x1 tiles and single-channel maps have the same bytes in both layouts)
        assert is_nhwc(blocks) == is_nhwc(image) or C == 1 or bs == 1, "packed and dense tensors must share the memory layout"
        Ck, Ek = (1, C * blocks.element_size()) if is_nhwc(image) else (C, blocks.element_size())   # channels-last: fat elements
        if n_exec > 0:
            with torch.cuda.device_of(blocks):
                self._arm(dyn)
                self._check(self.lib.bc_split(blocks.data_ptr(), image.data_ptr(), mapping_exec.data_ptr(), n_exec,
                                              N, Ck, H, W, bs, Ek, self._stream()), "split")
        return blocks

    def combine(self, blocks, out, grid_idx, mapping_exec, dyn=None):
        """out[executed tiles] <- blocks, in place.  reference: CombineFunction.forward, utils/block_funcs.py:87-124."""
        assert _ok(blocks) and _ok(out, blocks.dtype) and _ok(mapping_exec, torch.int32) and _ok(grid_idx, torch.int32)
        N, C, H, W = out.shape
        n_exec, Cb, bs, bs2 = blocks.shape
        _, one, GH, GW = grid_idx.shape
        assert bs >= 1 and bs == bs2 and one == 1 and grid_idx.size(0) == N and Cb == C
        assert GH * bs == H and GW * bs == W
        assert n_exec == mapping_exec.numel()
        assert is_nhwc(blocks) == is_nhwc(out) or C == 1 or bs == 1, "packed and dense tensors must share the memory layout"
        Ck, Ek = (1, C * blocks.element_size()) if is_nhwc(out) else (C, blocks.element_size())
        if n_exec > 0:
            with torch.cuda.device_of(blocks):
                self._arm(dyn)
                self._check(self.lib.bc_combine(blocks.data_ptr(), out.data_ptr(), mapping_exec.data_ptr(), n_exec,
                                                N, Ck, H, W, bs, Ek, self._stream()), "combine")
        return out

    def transfer(self, out, prev_computed, prev_transfer, prev_grid_idx, transfer_idx, padding):
        """border ring of non-executed tiles from the previous frame.  reference: TransferFunction.forward, :163-193."""
        assert _ok(out) and _ok(prev_computed, out.dtype) and _ok(prev_transfer, out.dtype) and _ok(transfer_idx, torch.int32)
        assert out.shape[1:] == prev_computed.shape[1:] == prev_transfer.shape[1:]
        assert not (is_nhwc(out) or is_nhwc(prev_computed) or is_nhwc(prev_transfer)), "the reference decomposition is NCHW only"
        N, _, GH, GW = prev_grid_idx.shape
        n_tr, C, bs, _ = out.shape
        assert n_tr == transfer_idx.numel()
        if n_tr > 0:
            with torch.cuda.device_of(out):
                self._check(self.lib.bc_transfer(out.data_ptr(),
                                                 prev_computed.data_ptr() if prev_computed.numel() else None,
                                                 prev_transfer.data_ptr() if prev_transfer.numel() else None,
                                                 transfer_idx.data_ptr(), n_tr, N, C, GH, GW, bs, int(padding),
                                                 out.element_size(), self._stream()), "transfer")
        return out

    def pad(self, data_exec, data_transfer, grid_idx, mapping_exec, pad):
        """halo-padded packed batch.  reference: BlockPadFunction.forward, utils/blockpad.py:23-71 (allocates the output)."""
        assert _ok(data_exec) and _ok(data_transfer, data_exec.dtype) and _ok(mapping_exec, torch.int32) and _ok(grid_idx, torch.int32)
        assert data_exec.shape[1:] == data_transfer.shape[1:], (data_exec.shape, data_transfer.shape)
        assert not (is_nhwc(data_exec) or is_nhwc(data_transfer)), "the reference decomposition is NCHW only"
        n_exec = mapping_exec.numel()
        assert n_exec <= data_exec.shape[0]
        assert grid_idx.numel() - n_exec <= data_transfer.shape[0], (grid_idx.numel(), n_exec, data_transfer.shape)
        assert pad > 0
        N, _, GH, GW = grid_idx.shape
        B, C, bs, _ = data_exec.shape
        out = torch.empty((B, C, bs + 2 * pad, bs + 2 * pad), device=data_exec.device, dtype=data_exec.dtype)
        if n_exec > 0:
            with torch.cuda.device_of(data_exec):
                self._check(self.lib.bc_pad(out.data_ptr(), data_exec.data_ptr(),
                                            data_transfer.data_ptr() if data_transfer.numel() else None,
                                            grid_idx.data_ptr(), mapping_exec.data_ptr(), n_exec, N, C, GH, GW, bs, int(pad),
                                            out.element_size(), self._stream()), "pad")
        return out

    # -- B. MI355X-first forms --------------------------------------------------------------------------
    def combine_copy(self, blocks, prev, out, grid_idx):
        """out <- blocks where executed else prev: fused scatter + copy, one pass over the dense map."""
        assert _ok(blocks) and _ok(prev, blocks.dtype) and _ok(out, blocks.dtype) and _ok(grid_idx, torch.int32)
        N, C, H, W = out.shape
        _, Cb, bs, _ = blocks.shape
        _, _, GH, GW = grid_idx.shape
        assert prev.shape == out.shape and Cb == C and GH * bs == H and GW * bs == W and grid_idx.size(0) == N
        assert out.data_ptr() != prev.data_ptr(), "combine_copy is out of place"
        assert is_nhwc(out) == is_nhwc(prev) and (is_nhwc(out) == is_nhwc(blocks) or C == 1 or bs == 1), \
            "blocks / prev / out must share the memory layout"
        Ck, Ek = (1, C * out.element_size()) if is_nhwc(out) else (C, out.element_size())
        with torch.cuda.device_of(out):
            self._check(self.lib.bc_combine_copy(blocks.data_ptr() if blocks.numel() else None, prev.data_ptr(), out.data_ptr(),
                                                 grid_idx.data_ptr(), N, Ck, H, W, bs, Ek, self._stream()),
                        "combine_copy")
        return out

    SLOT_WORDS = 3     # uint64 words of a bc_combine_copy_indirect slot buffer: prev, out, timing record

    def combine_copy_indirect(self, blocks, slots, grid_idx, out_shape, targets=None):
        """combine_copy as a hipGraph node: ``prev`` / ``out`` are read at run time from ``slots`` (device int64[3]: prev address,
        out address, timing record or 0; include/blockcopy_hip.h bc_combine_copy_indirect), so a captured launch serves a fresh
        output tensor every replay.  ``out_shape`` = (N,C,H,W) of the dense maps, which share the layout of ``blocks``;
        ``targets`` (the tensors behind the two addresses) is for checker backends only and ignored here."""
        assert _ok(blocks) and _ok(slots, torch.int64) and slots.numel() >= self.SLOT_WORDS and _ok(grid_idx, torch.int32)
        N, C, H, W = out_shape
        _, Cb, bs, _ = blocks.shape
        _, _, GH, GW = grid_idx.shape
        assert Cb == C and GH * bs == H and GW * bs == W and grid_idx.size(0) == N and blocks.numel() > 0
        Ck, Ek = (1, C * blocks.element_size()) if is_nhwc(blocks) else (C, blocks.element_size())
        with torch.cuda.device_of(blocks):
            # torch's caching allocator hands out 512-byte aligned blocks; the maps are whole allocations
            self._check(self.lib.bc_combine_copy_indirect(blocks.data_ptr(), slots.data_ptr(), grid_idx.data_ptr(), N, Ck, H, W, bs, Ek, 256,
                                                          self._stream()), "combine_copy_indirect")

    def combine_copy_cells(self, blocks, out_shape):
        """Timing cells (16 bytes each) one bc_combine_copy_indirect launch of this geometry writes when slot word 2 is set."""
        N, C, H, W = out_shape
        bs = blocks.shape[2]
        Ck, Ek = (1, C * blocks.element_size()) if is_nhwc(blocks) else (C, blocks.element_size())
        n = self.lib.bc_combine_copy_cells(blocks.data_ptr(), N, Ck, H, W, bs, Ek, 256)
        if n < 0:
            self._check(n, "combine_copy_cells")
        return n

    def tile_copy_indirect(self, dst, src_slot, mapping_exec, bs, n_exec_dev=None, target=None):
        """dst[executed tiles] <- the same tiles of the frame whose ADDRESS the device word ``src_slot`` (int64[1]) holds at run time:
        the network-input stage of a graph-replayed frame (the reference's split + combine_ of the input, core/blockcopy.py:62-68,
        without the packed tensor or a staging copy of the frame; include/blockcopy_hip.h bc_tile_copy_indirect).  ``dst`` (N,C,H,W)
        contiguous NCHW, same geometry as the source; ``n_exec_dev``: optional device int32 with this replay's executed-tile count;
        ``target`` (the tensor behind the address) is for checker backends only and ignored here."""
        assert _ok(dst) and dst.is_contiguous() and _ok(src_slot, torch.int64) and src_slot.numel() >= 1 and _ok(mapping_exec, torch.int32)
        assert n_exec_dev is None or (_ok(n_exec_dev, torch.int32) and n_exec_dev.numel() >= 1)
        N, C, H, W = dst.shape
        n_exec = mapping_exec.numel()
        if n_exec > 0:
            with torch.cuda.device_of(dst):
                # torch's caching allocator hands out 512-byte aligned blocks; a frame is a whole allocation or an aligned view of one
                self._check(self.lib.bc_tile_copy_indirect(dst.data_ptr(), src_slot.data_ptr(), mapping_exec.data_ptr(),
                                                           n_exec_dev.data_ptr() if n_exec_dev is not None else None, n_exec,
                                                           N, C, H, W, int(bs), dst.element_size(), 16, self._stream()), "tile_copy_indirect")
        return dst

    def pad_ring(self, data_exec, ring, grid_idx, mapping_exec, pad, prologue=None, dyn=None):
        """halo gather over the persistent ring cache (+ refresh of the executed tiles' rings).
        prologue = (scale, shift, relu): per-channel fp32 affine + ReLU fused into the gather (applied to
        values read from packed tiles; the ring keeps the ACTIVATED values, so records do not depend on the route that wrote them)."""
        assert _ok(data_exec) and _ok(ring, data_exec.dtype) and _ok(mapping_exec, torch.int32) and _ok(grid_idx, torch.int32)
        N, _, GH, GW = grid_idx.shape
        B, C, bs, _ = data_exec.shape
        n_exec = mapping_exec.numel()
        assert n_exec == B and pad > 0
        assert tuple(ring.shape) == (N * GH * GW, C, 4 * pad * bs), (ring.shape, (N * GH * GW, C, 4 * pad * bs))
        out = empty_like_layout((B, C, bs + 2 * pad, bs + 2 * pad), data_exec)
        if n_exec > 0 and is_nhwc(data_exec):
            scale, shift, relu = prologue if prologue is not None else (None, None, False)
            for v in (scale, shift):
                assert v is None or (_ok(v, torch.float32) and v.numel() == C)
            with torch.cuda.device_of(data_exec):
                self._arm(dyn)
                self._check(self.lib.bc_pad_ring_nhwc(out.data_ptr(), data_exec.data_ptr(), ring.data_ptr(), grid_idx.data_ptr(),
                                                      mapping_exec.data_ptr(), n_exec, N, C, GH, GW, bs, int(pad),
                                                      data_exec.element_size(), _DTYPE_CODE.get(data_exec.dtype, -1),
                                                      scale.data_ptr() if scale is not None else None,
                                                      shift.data_ptr() if shift is not None else None, int(bool(relu)),
                                                      self._stream()), "pad_ring_nhwc")
        elif n_exec > 0:
            with torch.cuda.device_of(data_exec):
                self._arm(dyn)
                if prologue is None:
                    self._check(self.lib.bc_pad_ring(out.data_ptr(), data_exec.data_ptr(), ring.data_ptr(), grid_idx.data_ptr(),
                                                     mapping_exec.data_ptr(), n_exec, N, C, GH, GW, bs, int(pad),
                                                     out.element_size(), self._stream()), "pad_ring")
                else:
                    scale, shift, relu = prologue
                    for v in (scale, shift):
                        assert v is None or (_ok(v, torch.float32) and v.numel() == C)
                    self._check(self.lib.bc_pad_ring_act(out.data_ptr(), data_exec.data_ptr(), ring.data_ptr(), grid_idx.data_ptr(),
                                                         mapping_exec.data_ptr(), n_exec, N, C, GH, GW, bs, int(pad),
                                                         _DTYPE_CODE[data_exec.dtype],
                                                         scale.data_ptr() if scale is not None else None,
                                                         shift.data_ptr() if shift is not None else None, int(bool(relu)),
                                                         self._stream()), "pad_ring_act")
        return out

    @staticmethod
    def pad_ring_add_supported(data_exec, add):
        return (is_nhwc(data_exec) and data_exec.dtype in _DTYPE_CODE and (data_exec.shape[1] * data_exec.element_size()) % 16 == 0
                and add.shape == data_exec.shape and add.dtype == data_exec.dtype and is_nhwc(add) and add.is_contiguous(memory_format=torch.channels_last))

    def pad_ring_add(self, data_exec, add, ring, grid_idx, mapping_exec, pad, prologue, dyn=None):
        """Halo gather of v = relu?(data*scale + shift + add) plus the plain v as a by-product: returns (padded, v).
        The ring cache of this op holds activated values (see include/blockcopy_hip.h)."""
        assert _ok(data_exec, *_DTYPE_CODE) and is_nhwc(data_exec) and _ok(add, data_exec.dtype) and _ok(ring, data_exec.dtype)
        assert _ok(mapping_exec, torch.int32) and _ok(grid_idx, torch.int32)
        N, _, GH, GW = grid_idx.shape
        B, C, bs, _ = data_exec.shape
        assert mapping_exec.numel() == B and add.shape == data_exec.shape and tuple(ring.shape) == (N * GH * GW, C, 4 * pad * bs)
        out = empty_like_layout((B, C, bs + 2 * pad, bs + 2 * pad), data_exec)
        act = empty_like_layout((B, C, bs, bs), data_exec)
        scale, shift, relu = prologue
        for v in (scale, shift):
            assert v is None or (_ok(v, torch.float32) and v.numel() == C)
        if B > 0:
            with torch.cuda.device_of(data_exec):
                self._arm(dyn)
                self._check(self.lib.bc_pad_ring_add_nhwc(out.data_ptr(), act.data_ptr(), data_exec.data_ptr(), add.data_ptr(), ring.data_ptr(),
                                                          grid_idx.data_ptr(), mapping_exec.data_ptr(), B, N, C, GH, GW, bs, int(pad),
                                                          _DTYPE_CODE[data_exec.dtype], scale.data_ptr() if scale is not None else None,
                                                          shift.data_ptr() if shift is not None else None, int(bool(relu)), self._stream()),
                            "pad_ring_add_nhwc")
        return out, act

    @staticmethod
    def maxpool3x3s2_supported(data_exec):
        bs = data_exec.shape[2]
        return (is_nhwc(data_exec) and data_exec.dtype in _DTYPE_CODE and bs == data_exec.shape[3] and bs >= 2 and bs % 2 == 0
                and (data_exec.shape[1] * data_exec.element_size()) % 16 == 0)

    def maxpool3x3s2_ring(self, data_exec, ring, grid_idx, mapping_exec, prologue=None, dyn=None):
        """Fused halo gather + max_pool2d(3, stride 2, padding 1) on a channels-last packed batch (ring as for pad_ring, pad 1)."""
        assert _ok(data_exec, *_DTYPE_CODE) and is_nhwc(data_exec) and _ok(ring, data_exec.dtype)
        assert _ok(mapping_exec, torch.int32) and _ok(grid_idx, torch.int32)
        N, _, GH, GW = grid_idx.shape
        B, C, bs, _ = data_exec.shape
        assert mapping_exec.numel() == B and tuple(ring.shape) == (N * GH * GW, C, 4 * bs)
        out = empty_like_layout((B, C, bs // 2, bs // 2), data_exec)
        scale, shift, relu = prologue if prologue is not None else (None, None, False)
        for v in (scale, shift):
            assert v is None or (_ok(v, torch.float32) and v.numel() == C)
        if B > 0:
            with torch.cuda.device_of(data_exec):
                self._arm(dyn)
                self._check(self.lib.bc_maxpool3x3s2_ring_nhwc(out.data_ptr(), data_exec.data_ptr(), ring.data_ptr(), grid_idx.data_ptr(),
                                                               mapping_exec.data_ptr(), B, N, C, GH, GW, bs, _DTYPE_CODE[data_exec.dtype],
                                                               scale.data_ptr() if scale is not None else None,
                                                               shift.data_ptr() if shift is not None else None, int(bool(relu)),
                                                               self._stream()), "maxpool3x3s2_ring_nhwc")
        return out

    @staticmethod
    def conv3x3_supported(data_exec, weight, stride=1, padding=1, dilation=1, groups=1):
        """Shapes the fused MFMA conv covers (everything else goes halo gather + library conv)."""
        def _one(v):
            return v if isinstance(v, int) else (v[0] if len(set(v)) == 1 else None)
        st = _one(stride)
        if st not in (1, 2) or data_exec.shape[2] % st:
            return False
        bs = data_exec.shape[2] // st                                 # output tile size
        cin_unit = 32 if data_exec.dtype == torch.float32 else 64     # 16-bit: two 32-channel units are staged per K iteration
        dil = _one(dilation)
        if dil == 2:        # the dilated stage of a detector backbone: padding = dilation = 2, stride 1, tiles of a multiple of 8 pixels
            return (data_exec.dtype in _DTYPE_CODE and weight.dtype == data_exec.dtype and is_nhwc(data_exec) and tuple(weight.shape[2:]) == (3, 3)
                    and _one(padding) == 2 and st == 1 and groups == 1 and weight.shape[1] % cin_unit == 0 and weight.shape[0] % 64 == 0
                    and data_exec.shape[2] == data_exec.shape[3] and bs % 8 == 0 and bs <= 248)
        return (data_exec.dtype in _DTYPE_CODE and weight.dtype == data_exec.dtype and is_nhwc(data_exec)
                and tuple(weight.shape[2:]) == (3, 3) and _one(padding) == 1 and dil == 1
                and groups == 1 and weight.shape[1] % cin_unit == 0 and weight.shape[0] % 64 == 0
                and data_exec.shape[2] == data_exec.shape[3]
                and (bs == 4 or (bs == 2 and data_exec.dtype != torch.bfloat16) or (bs % 8 == 0 and bs * st <= 248)))      # (2x2 tiles: fp16, fp32 in the split form)

    @staticmethod
    def pack_conv3x3_weights(weight):
        """(Cout, Cin, 3, 3) -> the MFMA operand stream of bc_conv3x3_ring_nhwc (include/blockcopy_hip.h), in the weight's dtype:
        wpk[nb][unit][tap][step][lane][j] = W[32*nb + lane%32][32*unit + 2*EPV*step + EPV*(lane//32) + j][tap] with EPV = elements
        per 16-byte vector (4 for fp32: 4 steps per 32-channel unit; 8 for fp16 / bf16: 2 steps)."""
        Cout, Cin, kh, kw = weight.shape
        assert (kh, kw) in ((3, 3), (1, 1)) and Cin % 32 == 0 and Cout % 32 == 0     # (1x1: the same stream with a single tap)
        epv = 16 // weight.element_size()
        steps = 32 // (2 * epv)
        w0 = weight.detach().as_subclass(torch.Tensor)
        w = w0.permute(2, 3, 1, 0).reshape(kh * kw, Cin // 32, steps, 2, epv, Cout // 32, 32)   # tap, unit, step, h, j, nb, n
        direct = w.permute(5, 1, 0, 2, 3, 6, 4).contiguous().view(-1)                                    # nb, unit, tap, step, h, n, j
        if w0.dtype != torch.float32:
            return direct
        # fp32: the SPLIT stream of the direct form on the 16-bit matrix pipe (codes | 0x2000, csrc/conv3x3_v2.inc BC_F32S) -- the direct stream
        # position by position, every 16-byte vector of four weights replaced by [hi0..3 | lo0..3] in fp16 with 16 w = hi + lo
        x16 = direct.view(-1, 4) * 16.0
        hi = x16.to(torch.float16)
        lo = (x16 - hi.float()).to(torch.float16)
        split = torch.cat([hi, lo], dim=1).contiguous().view(torch.float32).view(-1)
        if (kh, kw) != (3, 3):
            return torch.cat([direct, split])
        # fp32 3x3: the Winograd F(2x2,3x3) stream of csrc/conv3x3_wino.inc follows (16 values per (cin, cout)): U = G g Gt computed
        # in fp64, wino[nb16][chunk][step][q][lane = 16*kq + n][w] = U[f][32*chunk + 8*step + 2*kq + t][16*nb16 + n], 2*f + t = 4*q + w
        # G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1] applied along both filter axes as plain device arithmetic (no host constant: this may
        # run while a hipGraph is being captured, where a host-to-device copy is not permitted)
        def g_rows(t, dim):
            a, b, c = t.unbind(dim)
            return torch.stack([a, 0.5 * (a + b + c), 0.5 * (a - b + c), c], dim)
        U64 = g_rows(g_rows(w0.double(), 2), 3).permute(2, 3, 1, 0)                                          # xi, nu, cin, cout (fp64)
        U = U64.reshape(16, Cin, Cout).float()                                                              # f = 4*xi + nu, cin, cout
        U = U.reshape(16, Cin // 32, 4, 4, 2, Cout // 16, 16).permute(5, 1, 2, 0, 4, 3, 6)                  # nb, chunk, step, f, t, kq, n
        U16 = U.reshape(Cout // 16, Cin // 32, 4, 8, 4, 4, 16).permute(0, 1, 2, 3, 5, 6, 4)                 # nb, chunk, step, q, kq, n, w
        # third stream: the wide wave tile of csrc/conv3x3_wino32.inc (32 output channels per wave, 4-channel sub-steps):
        # wino32[nb32][chunk][ss][q][lane = 32*h + n][w] = U[f][32*chunk + 4*ss + 2*h + t][32*nb32 + n], 2*f + t = 4*q + w
        V = U64.reshape(8, 2, Cin // 32, 8, 2, 2, Cout // 32, 32).float()                                      # q, f%2, chunk, ss, h, t, nb, n
        V = V.permute(6, 2, 3, 0, 4, 7, 1, 5)                                                               # nb, chunk, ss, q, h, n, f%2, t
        # fourth stream: Winograd F(4x4,3x3) of csrc/conv3x3_wino4.inc (36 values per (cin, cout)), U = G g Gt in fp64 with
        # G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]:
        # wino4[cb][chunk][f = 6*xi + nu][lane = 16*kq + n][j] = U[f][16*chunk + 4*kq + j][16*cb + n]
        def g4_rows(t, dim):
            a, b, c = t.unbind(dim)
            return torch.stack([a / 4, -(a + b + c) / 6, -(a - b + c) / 6, a / 24 + b / 12 + c / 6, a / 24 - b / 12 + c / 6, c], dim)
        U4 = g4_rows(g4_rows(w0.double(), 2), 3).permute(2, 3, 1, 0).reshape(36, Cin, Cout).float()          # f, cin, cout
        U4 = U4.reshape(36, Cin // 16, 4, 4, Cout // 16, 16).permute(4, 1, 0, 2, 5, 3).contiguous()         # cb, chunk, f, kq, n, j
        # sixth stream: the F(4x4) stream position by position for its products on the 16-bit matrix pipe (codes 0x5000 | c, conv3x3_wino4.inc
        # SP): every 16-byte vector of four transformed weights replaced by [hi0..3 | lo0..3] in fp16 with 256 U = hi + lo
        u256 = U4.view(-1, 4) * 256.0
        uh = u256.to(torch.float16)
        ul = (u256 - uh.float()).to(torch.float16)
        u4_split = torch.cat([uh, ul], dim=1).contiguous().view(torch.float32).view(-1)
        return torch.cat([direct, U16.contiguous().view(-1), V.contiguous().view(-1), U4.view(-1), split, u4_split])

    # ---- pyramid pooling of a dense map in two launches (csrc/spp.inc): bc_spp_levels_nhwc + bc_spp_fuse_nhwc
    SPP_LDS_LIMIT = 150 * 1024      # both launchers refuse (BC_ERR_SHAPE) above this much dynamic LDS

    @classmethod
    def spp_supported(cls, x, co, n_levels, cout, grids=None):
        """x: ONE dense channels-last image (1, C, H, W).  With ``grids`` (the levels' (gh, gw) pairs) the launchers' LDS budgets are
        checked too -- bc_spp_levels_nhwc: partial sums + activated means + one level's weights; bc_spp_fuse_nhwc: two stages + the level
        maps + the block's folded BN + bilinear taps (csrc/blockcopy_hip.hip) -- so a module with the reference's default widths
        (bt_size 512, level_size 128: 268 KB) takes the generic route instead of failing inside the frame."""
        C = x.shape[1]
        if not (x.dim() == 4 and x.shape[0] == 1 and x.dtype in _DTYPE_CODE and x.is_contiguous(memory_format=torch.channels_last) and C % 4 == 0 and C <= 1024
                and 256 % (C // 4) == 0 and 1 <= n_levels <= 4 and 0 < co <= 1024 and cout % 64 == 0 and x.shape[2] * x.shape[3] * (C + n_levels * co) < 2 ** 31):
            return False
        if (((256 // (C // 4) + 1) * C + C * co) * 4) > cls.SPP_LDS_LIMIT:
            return False
        if grids is not None:
            n_bins = sum(int(gh) * int(gw) for gh, gw in grids)
            kp = (C + n_levels * co + 31) // 32 * 32
            if 2 * (64 * 9 + 2 * 256) * 16 + (n_bins * co + 2 * kp + 64 * 4 * 6) * 4 > cls.SPP_LDS_LIMIT:
                return False
        return True

    @staticmethod
    def pack_spp_level_weights(weights):
        """L weights (CO, C, 1, 1) -> fp32 [L][C][CO] (include/blockcopy_hip.h bc_spp_levels_nhwc)."""
        return torch.stack([w.detach().float().reshape(w.shape[0], w.shape[1]).t() for w in weights]).contiguous()

    def pack_spp_fuse_weights(self, weight):
        """(N, K, 1, 1) -> the fp32 one-tap stream of the weight zero-padded to a multiple of 32 input channels (bc_spp_fuse_nhwc)."""
        N, K = weight.shape[0], weight.shape[1]
        Kp = (K + 31) // 32 * 32
        w = torch.zeros((N, Kp, 1, 1), dtype=torch.float32, device=weight.device)
        w[:, :K] = weight.detach().float()
        return self.pack_conv3x3_weights(w)[:N * Kp].contiguous()      # (the direct one-tap stream only: no split stream behind it)

    def spp_levels(self, x, scale, shift, w, grids):
        """lv[bin][CO] of every level: conv1x1_l(relu(bn_l(adaptive_avg_pool2d(x, grid_l)))); bins of level l start at sum of the earlier grids.
        A batch of B maps (B, C, H, W) gives (B, bins, CO) in the same launch."""
        B, C, H, W = x.shape
        L, CO = w.shape[0], w.shape[2]
        assert tuple(w.shape) == (L, C, CO) and w.dtype == torch.float32 and w.is_contiguous() and len(grids) == L
        assert is_nhwc(x) or C == 1
        for v in (scale, shift):
            assert v is None or (_ok(v, torch.float32) and tuple(v.shape) == (L, C) and v.is_contiguous())
        n_bins = sum(gh * gw for gh, gw in grids)
        lv = torch.empty((B, n_bins, CO) if B > 1 else (n_bins, CO), dtype=x.dtype, device=x.device)
        garr = (ctypes.c_int * (2 * L))(*[int(v) for g in grids for v in g])
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device_of(x):
            self._check(self.lib.bc_spp_levels_n_nhwc(lv.data_ptr(), x.data_ptr(), ptr(scale), ptr(shift), w.data_ptr(), B, H, W, C, CO, L, garr,
                                                      _DTYPE_CODE[x.dtype], self._stream()), "spp_levels_nhwc")
        return lv

    def spp_fuse(self, x, lv, scale, shift, wpk, grids, cout, packed=None):
        """conv1x1_f(relu(bn_f(cat[x, upsampled levels]))) without the upsampled maps or the concatenation; returns (B, cout, H, W) channels-last.
        ``packed`` = (mapping_exec int32[n_exec], bs): ONE map, result (n_exec, cout, bs, bs) channels-last = the executed tiles only
        (bc_spp_fuse_packed_nhwc: what the gather of the dense result would give)."""
        B, C, H, W = x.shape
        if packed is not None:
            mapping_exec, bs = packed
            assert B == 1 and _ok(mapping_exec, torch.int32) and H % bs == 0 and W % bs == 0
            L, CO = len(grids), lv.shape[-1]
            K = C + L * CO
            assert lv.dtype == x.dtype and lv.is_contiguous() and wpk.dtype == torch.float32 and wpk.numel() == cout * ((K + 31) // 32 * 32)
            n_exec = mapping_exec.numel()
            out = torch.empty((n_exec, bs, bs, cout), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
            garr = (ctypes.c_int * (2 * L))(*[int(v) for g in grids for v in g])
            ptr = lambda t: t.data_ptr() if t is not None else None
            if n_exec > 0:
                with torch.cuda.device_of(x):
                    self._check(self.lib.bc_spp_fuse_packed_nhwc(out.data_ptr(), x.data_ptr(), lv.data_ptr(), ptr(scale), ptr(shift), wpk.data_ptr(),
                                                                 mapping_exec.data_ptr(), n_exec, int(bs), H, W, C, CO, L, garr, cout,
                                                                 _DTYPE_CODE[x.dtype], self._stream()), "spp_fuse_packed_nhwc")
            return out
        L, CO = len(grids), lv.shape[-1]
        K = C + L * CO
        for v in (scale, shift):
            assert v is None or (_ok(v, torch.float32) and v.numel() == K and v.is_contiguous())
        assert lv.dtype == x.dtype and lv.is_contiguous() and wpk.dtype == torch.float32 and wpk.numel() == cout * ((K + 31) // 32 * 32)
        assert lv.numel() == B * sum(gh * gw for gh, gw in grids) * CO
        out = torch.empty((B, H, W, cout), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
        garr = (ctypes.c_int * (2 * L))(*[int(v) for g in grids for v in g])
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device_of(x):
            self._check(self.lib.bc_spp_fuse_n_nhwc(out.data_ptr(), x.data_ptr(), lv.data_ptr(), ptr(scale), ptr(shift), wpk.data_ptr(), B, H, W, C, CO, L, garr,
                                                    cout, _DTYPE_CODE[x.dtype], self._stream()), "spp_fuse_nhwc")
        return out

    # ---- dense 3x3 conv to <= 4 output channels (detector prediction convs on the combined map): bc_pred3x3_nhwc
    @staticmethod
    def pred3x3_supported(x, weight, stride=1, padding=1, dilation=1, groups=1):
        def _one(v):
            return v if isinstance(v, (int, str)) else (v[0] if len(set(v)) == 1 else None)

        return (x.dim() == 4 and x.dtype in _DTYPE_CODE and x.is_contiguous(memory_format=torch.channels_last) and weight.dim() == 4
                and tuple(weight.shape[2:]) == (3, 3) and _one(stride) == 1 and _one(padding) == 1 and _one(dilation) == 1 and groups == 1
                and weight.shape[1] == x.shape[1] and x.shape[1] % 32 == 0 and 1 <= weight.shape[0] <= 4
                and x.numel() < 2 ** 31 and x.shape[0] <= 65535)

    @staticmethod
    def pack_pred3x3_weights(weight):
        """(Cout, Cin, 3, 3) -> fp32 [Cin][3][3][Cout] (the scalar operand order of k_pred3x3, include/blockcopy_hip.h)."""
        return weight.detach().float().permute(1, 2, 3, 0).contiguous()

    def pred3x3(self, x, wpk, bias, cout):
        """y = conv2d(x, w, bias, stride 1, padding 1) for a dense channels-last map and cout <= 4; returns (N, cout, H, W) channels-last."""
        N, C, H, W = x.shape
        assert x.is_contiguous(memory_format=torch.channels_last) and tuple(wpk.shape) == (C, 3, 3, cout) and wpk.dtype == torch.float32 and wpk.is_contiguous()
        assert bias is None or (bias.dtype == torch.float32 and bias.numel() == cout and bias.is_contiguous())
        out = torch.empty((N, H, W, cout), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
        with torch.cuda.device_of(x):
            self._check(self.lib.bc_pred3x3_nhwc(out.data_ptr(), x.data_ptr(), wpk.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                                 N, H, W, C, cout, _DTYPE_CODE[x.dtype], self._stream()), "pred3x3_nhwc")
        return out

    def conv3x3_candidates(self, n_exec, cin, cout, bs, elem_size=4, stride=1, dilation=1):
        """Decomposition indices of the balanced conv kernel that cover this layer shape (bs = input tile size), straight from
        the library's launcher rules (bc_conv3x3_candidates / bc_conv3x3_dil_candidates)."""
        buf = (ctypes.c_int * 128)()
        dt = 0 if elem_size == 4 else 1
        if dilation != 1:
            n = self.lib.bc_conv3x3_dil_candidates(dt, int(dilation), int(n_exec), int(cin), int(cout), int(bs), buf, 128)
        else:
            n = self.lib.bc_conv3x3_candidates(dt, int(stride), int(n_exec), int(cin), int(cout), int(bs), buf, 128)
        if n < 0:
            return []
        return [int(buf[k]) for k in range(n)]

    @staticmethod
    def time_routes(routes, reps=None, launches=4):
        """{name: median microseconds per call} of each callable, measured with events on the current stream.  A candidate the
        library rejects for this shape (BC_ERR_SHAPE: a decomposition that does not fit) is skipped; any other failure of a
        candidate is an error of the kernel, not a reason to pick another one, and propagates."""
        out = {}
        reps = reps or int(os.environ.get("BLOCKCOPY_CONV_TUNE_REPS", "3"))
        torch.cuda.synchronize()
        for name, fn in routes.items():
            try:
                fn()
                fn()
            except BlockCopyBackendError as e:
                if e.code != BC_ERR_SHAPE:
                    raise
                continue
            except RuntimeError:
                if name != "library":
                    raise
                continue      # the conv library's baseline could not run on this shape (no solver / out of memory): nothing to compare with, not an error of ours
            ts = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(launches):
                    fn()
                b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / launches)
            out[name] = sorted(ts)[len(ts) // 2]
        return out

    def conv3x3_ring(self, data_exec, ring, wpk, cout, grid_idx, mapping_exec, prologue=None, epilogue=None, cfg=None, stride=1, dilation=1, dyn=None):
        """Fused halo gather + 3x3/s1/p1 conv (+ optional epilogue) of a channels-last packed batch on the fp32 matrix
        cores.  prologue = (scale, shift, relu) per input channel, epilogue = (scale, shift, add, relu) per output channel."""
        dt = data_exec.dtype
        assert _ok(data_exec, *_DTYPE_CODE) and is_nhwc(data_exec) and _ok(ring, dt) and _ok(wpk, dt)
        assert _ok(mapping_exec, torch.int32) and _ok(grid_idx, torch.int32)
        N, _, GH, GW = grid_idx.shape
        B, C, bs, _ = data_exec.shape
        n_exec = mapping_exec.numel()
        assert n_exec == B and wpk.numel() in (9 * C * cout, 77 * C * cout, 86 * C * cout, 122 * C * cout)     # (fp32: direct + three Winograd streams + the two split streams)
        assert dilation in (1, 2) and (dilation == 1 or stride == 1)
        assert tuple(ring.shape) == (N * GH * GW, C, 4 * dilation * bs), (ring.shape, (N * GH * GW, C, 4 * dilation * bs))
        assert stride in (1, 2) and bs % stride == 0
        out = empty_like_layout((B, cout, bs // stride, bs // stride), data_exec)
        isc, ish, irelu = prologue if prologue is not None else (None, None, False)
        osc, osh, oadd, orelu = epilogue if epilogue is not None else (None, None, None, False)
        for v, n in ((isc, C), (ish, C), (osc, cout), (osh, cout)):
            assert v is None or (_ok(v, torch.float32) and v.numel() == n)
        assert oadd is None or (_ok(oadd, dt) and oadd.shape == out.shape and is_nhwc(oadd))
        ptr = lambda t: t.data_ptr() if t is not None else None
        if n_exec > 0:
            with torch.cuda.device_of(data_exec):
                want = int(cfg) if cfg is not None else self._conv_cfg_pinned      # explicit > pinned by tune() > library's choice
                if want != self._conv_cfg:
                    self._check(self.lib.bc_tune_set(b"conv2_cfg", want), "tune_set")
                    self._conv_cfg = want
                self._arm(dyn)
                if dilation == 2:
                    self._check(self.lib.bc_conv3x3_dil_ring_nhwc(out.data_ptr(), data_exec.data_ptr(), ring.data_ptr(), wpk.data_ptr(),
                                                                  grid_idx.data_ptr(), mapping_exec.data_ptr(), n_exec, N, C, cout, GH, GW, bs, 2,
                                                                  _DTYPE_CODE[data_exec.dtype], ptr(isc), ptr(ish), int(bool(irelu)),
                                                                  ptr(osc), ptr(osh), ptr(oadd), int(bool(orelu)), self._stream()),
                                "conv3x3_dil_ring_nhwc")
                    return out
                fn = self.lib.bc_conv3x3_ring_nhwc if stride == 1 else self.lib.bc_conv3x3s2_ring_nhwc
                self._check(fn(out.data_ptr(), data_exec.data_ptr(), ring.data_ptr(), wpk.data_ptr(),
                               grid_idx.data_ptr(), mapping_exec.data_ptr(), n_exec, N, C, cout, GH, GW,
                               bs, _DTYPE_CODE[data_exec.dtype], ptr(isc), ptr(ish), int(bool(irelu)),
                               ptr(osc), ptr(osh), ptr(oadd), int(bool(orelu)), self._stream()),
                            "conv3x3_ring_nhwc")
        return out

    # -- pointwise conv on the matrix cores (the fused kernel with one tap): prologue / epilogue fusion for 1x1 convs
    @staticmethod
    def conv1x1_geometry(data, stride):
        """(n_tiles, bs) the library is called with, or None: stride 1 = any 8x8 re-tiling of the pixels, stride 2 = the real tiles."""
        B, C, H, W = data.shape
        if stride == 1:
            return ((B * H * W) // 64, 8) if (B * H * W) % 64 == 0 else None
        if H != W or H % 2 or not ((H // 2) % 8 == 0 or H // 2 == 4 or (H // 2 == 2 and data.dtype != torch.bfloat16)) or H > 248:
            return None
        return B, H

    def conv1x1_supported(self, data, weight, stride=1, padding=0, dilation=1, groups=1):
        def _one(v):
            return v if isinstance(v, int) else (v[0] if len(set(v)) == 1 else None)
        st = _one(stride)
        if st not in (1, 2) or _one(padding) != 0 or _one(dilation) != 1 or groups != 1 or data.dim() != 4:
            return False
        cin_unit = 32 if data.dtype == torch.float32 else 64
        nhwc = is_nhwc(data) or (data.is_contiguous() and data.shape[2] * data.shape[3] == 1)
        return (data.is_cuda and data.dtype in _DTYPE_CODE and weight.dtype == data.dtype and nhwc and tuple(weight.shape[2:]) == (1, 1)
                and weight.shape[1] % cin_unit == 0 and weight.shape[0] % 64 == 0 and data.shape[1] == weight.shape[1]
                and self.conv1x1_geometry(data, st) is not None)

    def conv1x1_candidates(self, data, cout, stride=1):
        geo = self.conv1x1_geometry(data, stride)
        if geo is None:
            return []
        buf = (ctypes.c_int * 64)()
        n = self.lib.bc_conv1x1_candidates(_DTYPE_CODE[data.dtype], int(stride), geo[0], data.shape[1], int(cout), geo[1], buf, 64)
        return [int(buf[k]) for k in range(max(n, 0))]

    def conv1x1_upsample_supported(self, launch_kw, interp) -> bool:
        """Can the deferred pointwise conv ``launch_kw`` (the keyword arguments of conv1x1) carry "+ bilinear(interp[0])" in its epilogue
        (bc_conv_upsample_arm)?  Same packed tiles, power-of-two square output tiles, stride 1, a coarser map with the conv's output channels."""
        src, H, W, align, rh, rw = interp
        data, cout = launch_kw["data"], launch_kw["cout"]
        B, C, h, w = data.shape
        return (launch_kw.get("stride", 1) == 1 and data.is_cuda and src.is_cuda and src.dtype == data.dtype and src.dim() == 4
                and is_nhwc(src) and src.shape[0] == B and src.shape[1] == cout and src.shape[2] == src.shape[3]
                and (h, w) == (H, W) and H == W and H >= 2 and (H & (H - 1)) == 0 and self.conv1x1_geometry(data, 1) is not None)

    def conv1x1(self, data, wpk, cout, prologue=None, epilogue=None, cfg=None, stride=1, dyn=None, upsample=None):
        """relu?(conv1x1(prologue(data)) * scale + shift + add) of a channels-last tensor in one launch (bc_conv1x1_nhwc).
        ``upsample = (src, out_tile, align_corners, rh, rw)``: + bilinear(src) per tile before the add / ReLU (bc_conv_upsample_arm)."""
        assert _ok(data, *_DTYPE_CODE) and _ok(wpk, data.dtype)
        B, C, H, W = data.shape
        n_tiles, bs = self.conv1x1_geometry(data, stride)
        out = torch.empty((B, cout, H // stride, W // stride), dtype=data.dtype, device=data.device, memory_format=torch.channels_last)
        isc, ish, irelu = prologue if prologue is not None else (None, None, False)
        osc, osh, oadd, orelu = epilogue if epilogue is not None else (None, None, None, False)
        for v, n in ((isc, C), (ish, C), (osc, cout), (osh, cout)):
            assert v is None or (_ok(v, torch.float32) and v.numel() == n)
        if oadd is not None:
            oadd = oadd.contiguous(memory_format=torch.channels_last)
            assert oadd.dtype == data.dtype and oadd.shape == out.shape
        ptr = lambda t: t.data_ptr() if t is not None else None
        if out.numel() > 0:
            with torch.cuda.device_of(data):
                want = int(cfg) if cfg is not None else self._conv_cfg_pinned
                if want >= 0 and (want & 0x2000) and wpk.numel() < 2 * C * cout:
                    want &= ~0x2000      # (a buffer packed without the split stream: the fp32 pipe)
                if upsample is not None and want >= 0 and (want & 0x800):
                    want = -1       # (the GEMM form has no resampling epilogue: the library picks a direct decomposition)
                if want != self._conv_cfg:
                    self._check(self.lib.bc_tune_set(b"conv2_cfg", want), "tune_set")
                    self._conv_cfg = want
                self._arm(dyn)
                if upsample is not None:
                    usrc, out_bs, ualign, urh, urw = upsample
                    usrc = usrc.contiguous(memory_format=torch.channels_last)
                    assert stride == 1 and bs == 8 and usrc.dtype == data.dtype and usrc.shape[0] == B and usrc.shape[1] == cout
                    self._check(self.lib.bc_conv_upsample_arm(usrc.data_ptr(), int(usrc.shape[2]), int(out_bs), int(bool(ualign)),
                                                              float(urh), float(urw)), "conv_upsample_arm")
                self._check(self.lib.bc_conv1x1_nhwc(out.data_ptr(), data.data_ptr(), wpk.data_ptr(), n_tiles, C, cout, bs, int(stride),
                                                     _DTYPE_CODE[data.dtype], ptr(isc), ptr(ish), int(bool(irelu)), ptr(osc), ptr(osh),
                                                     ptr(oadd), int(bool(orelu)), self._stream()), "conv1x1_nhwc")
        return out

    # -- the network's last stage: activation prologue + pointwise conv to <= 32 channels (+ out-of-place combine), csrc/head1x1.inc
    @staticmethod
    def head1x1_supported(data, weight, stride=1, padding=0, dilation=1, groups=1):
        def _one(v):
            return v if isinstance(v, int) else (v[0] if len(set(v)) == 1 else None)
        if data.dim() != 4 or weight.dim() != 4 or _one(stride) != 1 or _one(padding) != 0 or _one(dilation) != 1 or groups != 1:
            return False
        cin_ok = (64, 128) if data.dtype == torch.float32 else (64, 128, 256)
        bs = data.shape[2]
        return (data.is_cuda and data.dtype in _DTYPE_CODE and weight.dtype == data.dtype and is_nhwc(data) and tuple(weight.shape[2:]) == (1, 1)
                and weight.shape[1] == data.shape[1] and data.shape[1] in cin_ok and 1 <= weight.shape[0] <= 32 and data.shape[2] == data.shape[3]
                and bs % 8 == 0 and ((bs >= 32 and bs % 32 == 0) or 32 % bs == 0))     # (not 24: the kernel stores 32-pixel blocks as runs of min(bs, 32))

    def pack_head1x1_weights(self, weight):
        """(Cout <= 32, Cin, 1, 1) -> the one-tap operand stream of bc_head1x1_scatter_nhwc: zero-padded to 32 output channels."""
        w = weight.detach().as_subclass(torch.Tensor)
        full = torch.zeros((32, w.shape[1], 1, 1), dtype=w.dtype, device=w.device)
        full[:w.shape[0]] = w
        return self.pack_conv3x3_weights(full)[:32 * w.shape[1]].contiguous()      # (the direct one-tap stream only)

    def _head_launch(self, out, data, wpk, cout, prologue, bias, scatter, grid_idx=None, mapping_exec=None, prev=None, slots=None, dyn=None):
        B, C, bs, _ = data.shape
        isc, ish, irelu = prologue if prologue is not None else (None, None, False)
        for v, n in ((isc, C), (ish, C), (bias, cout)):
            assert v is None or (_ok(v, torch.float32) and v.numel() == n)
        ptr = lambda t: t.data_ptr() if t is not None else None
        if scatter:
            N, _, GH, GW = grid_idx.shape
        else:
            N, GH, GW = 1, 1, max(B, 1)
        with torch.cuda.device_of(data):
            self._arm(dyn)
            self._check(self.lib.bc_head1x1_scatter_nhwc(ptr(out), data.data_ptr() if B else None, wpk.data_ptr(), ptr(prev), ptr(slots), ptr(grid_idx),
                                                         ptr(mapping_exec), B, N, C, cout, GH, GW, bs, _DTYPE_CODE[data.dtype], ptr(isc), ptr(ish),
                                                         int(bool(irelu)), ptr(bias), int(bool(scatter)), self._stream()), "head1x1_scatter_nhwc")

    def head1x1(self, data, wpk, cout, prologue=None, epilogue=None, cfg=None, stride=1, dyn=None):
        """Packed tiles (B, cout, bs, bs) channels-last = conv1x1(prologue(data)) [+ epilogue]; a per-channel shift (the conv bias) runs
        inside the kernel, anything else recorded after the conv as one bc_affine_act pass on the small result."""
        assert _ok(data, *_DTYPE_CODE) and is_nhwc(data) and _ok(wpk, data.dtype) and stride == 1
        B, C, bs, _ = data.shape
        out = torch.empty((B, cout, bs, bs), dtype=data.dtype, device=data.device, memory_format=torch.channels_last)
        osc, osh, oadd, orelu = epilogue if epilogue is not None else (None, None, None, False)
        bias_only = osc is None and oadd is None and not orelu
        if B > 0:
            self._head_launch(out, data, wpk, cout, prologue, osh if bias_only else None, False, dyn=dyn)
        if not bias_only:
            out = self.affine_act(out, osc, osh, oadd, orelu, dyn=dyn)
        return out

    def head1x1_scatter(self, data, wpk, cout, prologue, bias, grid_idx, mapping_exec, prev=None, out=None, slots=None, targets=None, dyn=None):
        """The fresh dense map (N, cout, H, W) channels-last <- executed tiles conv1x1(prologue(data)) + bias at their grid positions,
        skipped tiles from ``prev``.  Either ``out`` (+ ``prev`` unless every tile is executed) or ``slots`` (hipGraph node: device
        int64 words [prev address, out address, ...]; ``targets`` = the tensors behind them, for checker backends only)."""
        assert _ok(data, *_DTYPE_CODE) and is_nhwc(data) and _ok(wpk, data.dtype) and _ok(grid_idx, torch.int32) and _ok(mapping_exec, torch.int32)
        B, C, bs, _ = data.shape
        N, _, GH, GW = grid_idx.shape
        assert mapping_exec.numel() == B
        if slots is None:
            assert out is not None and tuple(out.shape) == (N, cout, GH * bs, GW * bs) and out.dtype == data.dtype and (is_nhwc(out) or cout == 1)
            assert prev is None or (prev.shape == out.shape and prev.dtype == out.dtype and is_nhwc(prev) == is_nhwc(out) and prev.data_ptr() != out.data_ptr())
            assert prev is not None or B == N * GH * GW, "skipped tiles need the previous frame's map"
        else:
            assert _ok(slots, torch.int64) and slots.numel() >= 2
        self._head_launch(out if slots is None else None, data, wpk, cout, prologue, bias, True, grid_idx, mapping_exec, prev if slots is None else None, slots, dyn=dyn)
        return out

    # -- adaptive average pooling of dense channels-last maps (pyramid pooling)
    @staticmethod
    def adaptive_avg_pool_supported(x):
        if not (x.is_cuda and x.dim() == 4 and x.dtype in _DTYPE_CODE and is_nhwc(x) and 0 < x.numel() < 2 ** 31):
            return False
        kv = x.shape[1] * x.element_size()
        return kv % 16 == 0 and kv // 16 <= 256 and 256 % (kv // 16) == 0

    def adaptive_avg_pool(self, x, out_hw):
        """F.adaptive_avg_pool2d(x, out_hw) of a channels-last tensor (bc_adaptive_avg_pool_nhwc)."""
        assert self.adaptive_avg_pool_supported(x)
        N, C, H, W = x.shape
        OH, OW = int(out_hw[0]), int(out_hw[1])
        out = torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        with torch.cuda.device_of(x):
            self._check(self.lib.bc_adaptive_avg_pool_nhwc(out.data_ptr(), x.data_ptr(), N, C, H, W, OH, OW, _DTYPE_CODE[x.dtype], self._stream()),
                        "adaptive_avg_pool_nhwc")
        return out

    # -- training-mode BatchNorm forward of the policy net (two launches instead of five)
    @staticmethod
    def bn_train_supported(x):
        if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and 0 < x.numel() < 2 ** 32 and x.shape[0] * x.shape[2] * x.shape[3] < 2 ** 31):
            return False
        if x.is_contiguous():
            return True
        C = x.shape[1]
        return is_nhwc(x) and C % 4 == 0 and C // 4 <= 256 and 256 % (C // 4) == 0

    def bn_train(self, x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu=False):
        """(y, save_mean, save_invstd) of training-mode batch_norm(+ReLU) on fp32 ``x`` (NCHW: bc_bn_train_fwd; channels-last:
        bc_bn_train_stats_nhwc + bc_affine_act_nhwc); running statistics and the batch counter are updated in place."""
        assert self.bn_train_supported(x)
        N, C, H, W = x.shape
        dev = x.device
        save_mean, save_invstd = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
        for v in (weight, bias, running_mean, running_var):
            assert v is None or (_ok(v, torch.float32) and v.numel() == C)
        assert num_batches_tracked is None or (num_batches_tracked.dtype == torch.int64 and num_batches_tracked.is_cuda)
        ptr = lambda t: t.data_ptr() if t is not None else None
        if x.is_contiguous():
            y = torch.empty_like(x)
            ws = torch.empty(C * 64 * 2, dtype=torch.float32, device=dev)
            with torch.cuda.device_of(x):
                self._check(self.lib.bc_bn_train_fwd(y.data_ptr(), x.data_ptr(), N, C, H * W, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var),
                                                     ptr(num_batches_tracked), save_mean.data_ptr(), save_invstd.data_ptr(), float(momentum), float(eps),
                                                     int(bool(relu)), ws.data_ptr(), ws.numel(), self._stream()), "bn_train_fwd")
            return y, save_mean, save_invstd
        scale, shift = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
        ws = torch.empty(512 * C * 2, dtype=torch.float32, device=dev)
        with torch.cuda.device_of(x):
            self._check(self.lib.bc_bn_train_stats_nhwc(x.data_ptr(), N * H * W, C, _DTYPE_CODE[x.dtype], float(eps), ptr(weight), ptr(bias), ptr(running_mean),
                                                        ptr(running_var), ptr(num_batches_tracked), float(momentum), save_mean.data_ptr(), save_invstd.data_ptr(),
                                                        scale.data_ptr(), shift.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), "bn_train_stats_nhwc")
        return self.affine_act(x, scale, shift, None, relu), save_mean, save_invstd

    # -- L2 normalisation of every pixel over its channels, scaled, written into a channel slice of a wider tensor (bc_l2norm_cat_nhwc)
    @staticmethod
    def l2norm_cat_supported(x, out=None):
        if not (x.is_cuda and x.dim() == 4 and x.dtype in _DTYPE_CODE and x.numel() > 0 and is_nhwc(x)):
            return False
        ve = 16 // x.element_size()
        return x.shape[1] % ve == 0 and x.shape[1] // ve <= 256

    def l2norm_cat(self, out, c_off, x, weight, eps):
        """out[:, c_off : c_off + C] = weight[c] * x / (||x||_2 over channels + eps), channels-last tensors of one spatial shape."""
        B, C, H, W = x.shape
        assert self.l2norm_cat_supported(x) and out.dtype == x.dtype and is_nhwc(out) and out.shape[0] == B and tuple(out.shape[2:]) == (H, W)
        assert _ok(weight, torch.float32) and weight.numel() == C and 0 <= c_off and c_off + C <= out.shape[1]
        with torch.cuda.device_of(x):
            self._check(self.lib.bc_l2norm_cat_nhwc(out.data_ptr(), x.data_ptr(), weight.data_ptr(), B * H * W, C, out.shape[1], int(c_off), float(eps),
                                                    _DTYPE_CODE[x.dtype], self._stream()), "l2norm_cat_nhwc")
        return out

    @staticmethod
    def pack_deconv4_weights(weight):
        """ConvTranspose2d weight (Cin, Cout, 4, 4) -> the pointwise conv (16 Cout, Cin, 1, 1) whose output channel (4 ky + kx) Cout + co is
        tap (ky, kx) of output channel co, packed for bc_conv1x1_nhwc (l2norm_cat_deconv gathers the taps)."""
        cin, cout, kh, kw = weight.shape
        assert (kh, kw) == (4, 4)
        w = weight.detach().as_subclass(torch.Tensor).permute(2, 3, 1, 0).reshape(16 * cout, cin, 1, 1).contiguous()
        return HipBackend.pack_conv3x3_weights(w)

    def l2norm_cat_deconv(self, out, c_off, t, bias, weight, stride, eps):
        """out[:, c_off : c_off + C] = L2Norm(conv_transpose(k4, stride 4 pad 0 | stride 2 pad 1) + bias) from the 16-tap patches t (B, 16 C, h, w)
        (channels-last) of a pointwise conv with pack_deconv4_weights."""
        B, C16, h, w = t.shape
        C = C16 // 16
        assert is_nhwc(t) and is_nhwc(out) and out.dtype == t.dtype and out.shape[0] == B and tuple(out.shape[2:]) == (h * stride, w * stride)
        assert _ok(weight, torch.float32) and weight.numel() == C and (bias is None or (_ok(bias, torch.float32) and bias.numel() == C))
        with torch.cuda.device_of(t):
            self._check(self.lib.bc_l2norm_cat_deconv_nhwc(out.data_ptr(), t.data_ptr(), bias.data_ptr() if bias is not None else None, weight.data_ptr(),
                                                           B, h, w, C, out.shape[1], int(c_off), int(stride), float(eps), _DTYPE_CODE[t.dtype],
                                                           self._stream()), "l2norm_cat_deconv_nhwc")
        return out

    # -- group_norm over all executed tiles as a per-channel affine map (one read of the tensor)
    @staticmethod
    def group_norm_affine_supported(data, groups):
        if not (data.is_cuda and data.dim() == 4 and data.dtype in _DTYPE_CODE and data.numel() > 0):
            return False
        C = data.shape[1]
        if not (is_nhwc(data) or (data.is_contiguous() and (C == 1 or data.shape[2] * data.shape[3] == 1))):
            return False
        ve = 16 // data.element_size()
        return C % groups == 0 and C % ve == 0 and C // ve <= 256 and 256 % (C // ve) == 0

    def group_norm_affine(self, data, groups, weight=None, bias=None, eps=1e-5):
        """(scale, shift), float32[C]: group_norm of the channels-last packed tiles ``data`` (statistics over all tiles, the
        reference's batched form, core/tensorwrapper.py:600-633) == data * scale[c] + shift[c]  (bc_group_norm_affine_nhwc)."""
        assert self.group_norm_affine_supported(data, groups)
        B, C, H, W = data.shape
        dev = data.device
        scale, shift = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
        ws = torch.empty(512 * C * 2, dtype=torch.float32, device=dev)
        for v in (weight, bias):
            assert v is None or (_ok(v, torch.float32) and v.numel() == C)
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device_of(data):
            self._check(self.lib.bc_group_norm_affine_nhwc(data.data_ptr(), B * H * W, C, int(groups), _DTYPE_CODE[data.dtype], float(eps), ptr(weight),
                                                           ptr(bias), scale.data_ptr(), shift.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                        "group_norm_affine_nhwc")
        return scale, shift

    # -- network-input stage: window gather from the frame-state map + 7x7 / stride 2 stem conv (csrc/stem7x7.inc)
    @staticmethod
    def stem7x7_supported(frame_state, weight, bs, stride=2, padding=3, dilation=1, groups=1):
        def _one(v):
            return v if isinstance(v, int) else (v[0] if len(set(v)) == 1 else None)
        return (frame_state.is_cuda and frame_state.dim() == 4 and frame_state.shape[1] == 3 and frame_state.is_contiguous()
                and frame_state.dtype in _DTYPE_CODE and weight.dtype == frame_state.dtype and tuple(weight.shape) == (64, 3, 7, 7)
                and _one(stride) == 2 and _one(padding) == 3 and _one(dilation) == 1 and groups == 1 and bs % 64 == 0
                and frame_state.shape[2] % bs == 0 and frame_state.shape[3] % bs == 0)

    @staticmethod
    def pack_stem7x7_weights(weight):
        """(64, 3, 7, 7) -> the operand stream of bc_stem7x7s2_nhwc (include/blockcopy_hip.h): per output-row tap ky the 21 (kx, c)
        values as one K segment, zero-padded to 24 (fp32) / 32 (16-bit), in MFMA lane order."""
        w = weight.detach().as_subclass(torch.Tensor)
        assert tuple(w.shape) == (64, 3, 7, 7)
        kseg = 24 if w.element_size() == 4 else 32
        seg = torch.zeros((64, 7, kseg), dtype=w.dtype, device=w.device)
        seg[:, :, :21] = w.permute(0, 2, 3, 1).reshape(64, 7, 21)                 # [co][ky][3*kx + c]
        if w.element_size() == 4:
            v = seg.reshape(2, 32, 7, 3, 4, 2).permute(0, 2, 3, 5, 1, 4)            # nb, n, ky, t4, j, h -> nb, ky, t4, h, n, j
            # + the SPLIT streams (bc_tune "stem_split": the fp32 frame on the 16-bit matrix pipe): 16 w = hi + lo in fp16, each in the 16-bit order
            seg16 = torch.zeros((64, 7, 32), dtype=torch.float32, device=w.device)
            seg16[:, :, :21] = seg[:, :, :21] * 16.0
            hi = seg16.to(torch.float16)
            lo = (seg16 - hi.float()).to(torch.float16)
            order16 = lambda t: t.reshape(2, 32, 7, 2, 2, 8).permute(0, 2, 3, 4, 1, 5).contiguous().view(-1).view(torch.float32)
            return torch.cat([v.contiguous().view(-1), order16(hi), order16(lo)])
        v = seg.reshape(2, 32, 7, 2, 2, 8).permute(0, 2, 3, 4, 1, 5)                # nb, n, ky, s, h, j  -> nb, ky, s, h, n, j
        return v.contiguous().view(-1)

    def stem7x7(self, frame_state, wpk, mapping_exec, bs, epilogue=None, dyn=None):
        """(n_exec, 64, bs/2, bs/2) channels-last = epilogue(conv7x7 s2 p3 of the (bs+6)^2 windows of ``frame_state``)."""
        assert _ok(frame_state, *_DTYPE_CODE) and frame_state.is_contiguous() and _ok(wpk, frame_state.dtype) and _ok(mapping_exec, torch.int32)
        N, C, H, W = frame_state.shape
        n_exec = mapping_exec.numel()
        out = torch.empty((n_exec, 64, bs // 2, bs // 2), dtype=frame_state.dtype, device=frame_state.device, memory_format=torch.channels_last)
        osc, osh, oadd, orelu = epilogue if epilogue is not None else (None, None, None, False)
        for v in (osc, osh):
            assert v is None or (_ok(v, torch.float32) and v.numel() == 64)
        assert oadd is None or (_ok(oadd, frame_state.dtype) and oadd.shape == out.shape and is_nhwc(oadd))
        ptr = lambda t: t.data_ptr() if t is not None else None
        if n_exec > 0:
            with torch.cuda.device_of(frame_state):
                self._arm(dyn)
                self._check(self.lib.bc_stem7x7s2_nhwc(out.data_ptr(), frame_state.data_ptr(), wpk.data_ptr(), mapping_exec.data_ptr(), n_exec,
                                                       N, H, W, int(bs), 64, _DTYPE_CODE[frame_state.dtype], ptr(osc), ptr(osh), ptr(oadd),
                                                       int(bool(orelu)), self._stream()), "stem7x7s2_nhwc")
        return out

    def affine_act(self, data, scale=None, shift=None, add=None, relu=False, dyn=None):
        """relu?(data*scale[c] + shift[c] + add) on a packed (B,C,h,w) tensor in one pass (fp32 arithmetic)."""
        assert _ok(data, *_DTYPE_CODE) and data.dim() == 4
        B, C, h, w = data.shape
        for v in (scale, shift):
            assert v is None or (_ok(v, torch.float32) and v.numel() == C)
        assert add is None or (_ok(add, data.dtype) and add.shape == data.shape)
        out = torch.empty_like(data)   # preserves the memory format
        if add is not None and is_nhwc(add) != is_nhwc(data):
            add = add.contiguous(memory_format=torch.channels_last if is_nhwc(data) else torch.contiguous_format)
        if data.numel() > 0 and is_nhwc(data):
            with torch.cuda.device_of(data):
                self._arm(dyn)
                self._check(self.lib.bc_affine_act_nhwc(out.data_ptr(), data.data_ptr(), add.data_ptr() if add is not None else None,
                                                        scale.data_ptr() if scale is not None else None,
                                                        shift.data_ptr() if shift is not None else None, int(bool(relu)),
                                                        B * h * w, C, _DTYPE_CODE[data.dtype], self._stream()), "affine_act_nhwc")
        elif data.numel() > 0:
            with torch.cuda.device_of(data):
                self._check(self.lib.bc_affine_act(out.data_ptr(), data.data_ptr(), add.data_ptr() if add is not None else None,
                                                   scale.data_ptr() if scale is not None else None,
                                                   shift.data_ptr() if shift is not None else None, int(bool(relu)),
                                                   B, C, h * w, _DTYPE_CODE[data.dtype], self._stream()), "affine_act")
        return out

    supports_fusion_dtypes = tuple(_DTYPE_CODE)

    @staticmethod
    def interp_epilogue_supported(data):
        """Deferred interpolation (epilogue fused into the resampling launch) exists for channels-last tensors."""
        return data.dim() == 4 and is_nhwc(data) and data.dtype in _DTYPE_CODE

    def upsample_argmax(self, logits, size, align_corners=False):
        """int64 (N, H, W): arg-max over dim 1 of F.interpolate(logits, size, mode='bilinear', align_corners=...), without the
        (N, C, H, W) intermediate (include/blockcopy_hip.h bc_upsample_argmax).  Any dense layout of ``logits``."""
        assert _ok(logits, *_DTYPE_CODE) and logits.dim() == 4
        N, C, h, w = logits.shape
        H, W = int(size[0]), int(size[1])
        out = torch.empty((N, H, W), dtype=torch.int64, device=logits.device)
        f32 = np.float32                   # (ATen's area_pixel_compute_scale of a `size=` call, in float32)
        if align_corners:
            rh, rw = (f32(h - 1) / f32(H - 1) if H > 1 else f32(0)), (f32(w - 1) / f32(W - 1) if W > 1 else f32(0))
        else:
            rh, rw = f32(h) / f32(H), f32(w) / f32(W)
        sn, sc, sy, sx = logits.stride()
        if out.numel() > 0:
            with torch.cuda.device_of(logits):
                self._check(self.lib.bc_upsample_argmax(out.data_ptr(), logits.data_ptr(), N, C, h, w, H, W, sn, sc, sy, sx, int(bool(align_corners)),
                                                        float(rh), float(rw), _DTYPE_CODE[logits.dtype], self._stream()),
                            "upsample_argmax")
        return out

    def interp_bilinear(self, data, out_h, out_w, align_corners, rh, rw, epilogue=None, dyn=None):
        """per-tile bilinear resampling (planes = all leading dims); arithmetic of torch's upsample_bilinear2d.
        epilogue = (scale, shift, add, relu) on the resampled value (channels-last only)."""
        assert _ok(data, *_DTYPE_CODE) and data.dim() == 4
        B, C, h, w = data.shape
        out = empty_like_layout((B, C, out_h, out_w), data)
        if epilogue is not None:
            assert is_nhwc(data), "interp epilogue: channels-last only"
            scale, shift, add, relu = epilogue
            for v in (scale, shift):
                assert v is None or (_ok(v, torch.float32) and v.numel() == C)
            assert add is None or (_ok(add, data.dtype) and tuple(add.shape) == tuple(out.shape) and is_nhwc(add))
            if out.numel() > 0:
                with torch.cuda.device_of(data):
                    self._arm(dyn)
                    self._check(self.lib.bc_interp_bilinear_act_nhwc(out.data_ptr(), data.data_ptr(), B, C, h, w, out_h, out_w,
                                                                     int(bool(align_corners)), float(rh), float(rw), _DTYPE_CODE[data.dtype],
                                                                     scale.data_ptr() if scale is not None else None,
                                                                     shift.data_ptr() if shift is not None else None,
                                                                     add.data_ptr() if add is not None else None, int(bool(relu)),
                                                                     self._stream()), "interp_bilinear_act_nhwc")
            return out
        if out.numel() > 0 and is_nhwc(data):
            with torch.cuda.device_of(data):
                self._arm(dyn)
                self._check(self.lib.bc_interp_bilinear_nhwc(out.data_ptr(), data.data_ptr(), B, C, h, w, out_h, out_w,
                                                             int(bool(align_corners)), float(rh), float(rw),
                                                             _DTYPE_CODE[data.dtype], self._stream()), "interp_bilinear_nhwc")
        elif out.numel() > 0:
            with torch.cuda.device_of(data):
                self._check(self.lib.bc_interp_bilinear(out.data_ptr(), data.data_ptr(), B * C, h, w, out_h, out_w,
                                                        int(bool(align_corners)), float(rh), float(rw),
                                                        _DTYPE_CODE[data.dtype], self._stream()), "interp_bilinear")
        return out

    supports_interp_dtypes = tuple(_DTYPE_CODE)

    def grid_tables_device(self, grid, prev_grid_idx=None):
        """index tables computed on the GPU (no host sync).  Returns (grid_idx, mapping_buf, transfer_buf, counts)
        where mapping_buf / transfer_buf are full-length buffers whose valid prefix lengths are counts[0] / counts[1]."""
        assert _ok(grid, torch.bool, torch.uint8)
        n_total = grid.numel()
        grid_idx = torch.empty(grid.shape, dtype=torch.int32, device=grid.device)
        mapping = torch.empty(n_total, dtype=torch.int32, device=grid.device)
        counts = torch.empty(2, dtype=torch.int32, device=grid.device)
        transfer = torch.empty(n_total, dtype=torch.int32, device=grid.device) if prev_grid_idx is not None else None
        with torch.cuda.device_of(grid):
            self._check(self.lib.bc_grid_tables(grid.data_ptr(), n_total, grid_idx.data_ptr(), mapping.data_ptr(),
                                                prev_grid_idx.data_ptr() if prev_grid_idx is not None else None,
                                                transfer.data_ptr() if transfer is not None else None,
                                                counts.data_ptr(), self._stream()), "grid_tables")
        return grid_idx, mapping, transfer, counts

    def policy_step(self, logits, seed: int, counter: int, multiple: int, at_least_one: bool, grid_u8, tables, counts, mailbox=None):
        """Device policy decision (include/blockcopy_hip.h bc_policy_step): Bernoulli(logits) + round-up-to-multiple + index
        tables in one launch.  ``grid_u8`` uint8[n_total], ``tables`` int32[2*n_total] = [grid_idx | mapping_exec], ``counts``
        int32[4] are caller-owned device buffers; ``mailbox`` an optional pinned-host (or device) int32[4] that receives the counts as well."""
        assert _ok(logits, torch.float32) and logits.is_contiguous() and _ok(grid_u8, torch.uint8) and _ok(tables, torch.int32) and _ok(counts, torch.int32)
        n_total = logits.numel()
        assert grid_u8.numel() == n_total and tables.numel() == 2 * n_total and counts.numel() >= 4
        assert mailbox is None or ((mailbox.is_pinned() or mailbox.is_cuda) and mailbox.dtype == torch.int32 and mailbox.numel() >= 4 and mailbox.is_contiguous())
        with torch.cuda.device_of(logits):
            self._check(self.lib.bc_policy_step(logits.data_ptr(), n_total, int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1),
                                                int(multiple), int(bool(at_least_one)), grid_u8.data_ptr(), tables.data_ptr(),
                                                tables.data_ptr() + 4 * n_total, counts.data_ptr(),
                                                mailbox.data_ptr() if mailbox is not None else None, self._stream()), "policy_step")

    def policy_features(self, sources, h: int, w: int):
        """One-gather input of the policy net: ``sources`` = 4 x (tensor (N,C,H,W) any strides, scale_h, scale_w, offset) ->
        float32 (N, sum C, h, w) = concat of nearest-resampled sources (+ offset), ATen 'nearest' index arithmetic."""
        assert len(sources) == 4
        N = sources[0][0].shape[0]
        ptrs = (ctypes.c_void_p * 4)()
        strides = (ctypes.c_longlong * 16)()
        dims = (ctypes.c_int * 16)()
        scales = (ctypes.c_float * 12)()
        ctot = 0
        for k, (t, sh, sw, off) in enumerate(sources):
            assert t.is_cuda and t.dim() == 4 and t.shape[0] == N
            code = 3 if t.dtype in (torch.bool, torch.uint8) else _DTYPE_CODE[t.dtype]
            ptrs[k] = t.data_ptr()
            strides[4 * k:4 * k + 4] = list(t.stride())
            dims[4 * k:4 * k + 4] = [t.shape[1], t.shape[2], t.shape[3], code]
            scales[3 * k:3 * k + 3] = [float(sh), float(sw), float(off)]
            ctot += t.shape[1]
        out = torch.empty((N, ctot, h, w), dtype=torch.float32, device=sources[0][0].device)
        with torch.cuda.device_of(out):
            self._check(self.lib.bc_policy_features(out.data_ptr(), N, h, w, ptrs, strides, dims, scales, self._stream()), "policy_features")
        self._keepalive = (ptrs, strides, dims, scales)   # (arguments are read at enqueue time; kept for tidiness)
        return out

    def grid_tables_host(self, grid_u8: np.ndarray, grid_idx: np.ndarray, mapping: np.ndarray,
                         prev_grid_idx: np.ndarray = None, transfer: np.ndarray = None) -> int:
        """host-side tables (C loop in the library) written straight into caller-provided (pinned) int32 arrays."""
        rc = self.lib.bc_grid_tables_host(grid_u8.ctypes.data, grid_u8.size, grid_idx.ctypes.data, mapping.ctypes.data,
                                          prev_grid_idx.ctypes.data if prev_grid_idx is not None else None,
                                          transfer.ctypes.data if transfer is not None else None)
        if rc < 0:
            self._check(rc, "grid_tables_host")
        return rc

    def nms(self, dets, iou_thr):
        """Greedy NMS on the device (reference: mmdet.ops.nms.nms_wrapper.nms -> nms_cuda).  dets: (n,5) float32
        [x1,y1,x2,y2,score].  Returns (kept dets, original indices of the kept boxes ascending) -- the reference's
        return contract; the index count needs one D->H read (shapes), everything else stays on the GPU."""
        assert _ok(dets, torch.float32) and dets.dim() == 2 and dets.shape[1] == 5
        n = dets.shape[0]
        if n == 0:
            return dets, torch.zeros(0, dtype=torch.long, device=dets.device)
        assert n <= 4096, "bc_nms_sorted handles up to 4096 boxes (the CSP head keeps nms_pre = 1000)"
        order = dets[:, 4].sort(0, descending=True)[1]
        srt = dets.index_select(0, order).contiguous()
        cb = (n + 63) // 64
        ws = torch.empty(n * cb + n, dtype=torch.int64, device=dets.device)      # suppression words + per-box in-block suppressors
        keep = torch.empty(n, dtype=torch.int32, device=dets.device)
        count = torch.empty(1, dtype=torch.int32, device=dets.device)
        with torch.cuda.device_of(dets):
            self._check(self.lib.bc_nms_sorted(srt.data_ptr(), n, float(iou_thr), ws.data_ptr(), keep.data_ptr(), count.data_ptr(),
                                               self._stream()), "nms_sorted")
        k = int(count.item())
        inds = order[keep[:k].long()].sort(0)[0]
        return dets[inds, :], inds

    def csp_decode_nms(self, scores, top, heights, off_y, off_x, map_w, stride, wh_ratio, img_shape, score_thr, iou_thr, max_out):
        """The detector decode between top-k and the kept boxes without a host round trip in between (bc_csp_decode +
        bc_nms_sorted_dev): ``scores`` / ``top`` = the output of ``topk`` on the score map (sorted descending, flat positions), ``heights`` /
        ``off_y`` / ``off_x`` gathered at ``top``.  Returns the kept boxes (k, 5) in descending score order, at most ``max_out``; the
        only D->H read is the kept count."""
        assert _ok(scores, torch.float32) and _ok(top, torch.int64) and _ok(heights, torch.float32) and _ok(off_y, torch.float32) and _ok(off_x, torch.float32)
        k = scores.numel()
        assert 0 < k <= 4096 and top.numel() == k and heights.numel() == k and off_y.numel() == k and off_x.numel() == k
        dev = scores.device
        dets = torch.empty((k, 5), dtype=torch.float32, device=dev)
        cnt = torch.empty(2, dtype=torch.int32, device=dev)           # [selected, kept]
        ws = torch.empty(k * ((k + 63) // 64) + k, dtype=torch.int64, device=dev)
        keep = torch.empty(k, dtype=torch.int32, device=dev)
        with torch.cuda.device_of(scores):
            self._check(self.lib.bc_csp_decode(scores.contiguous().data_ptr(), top.contiguous().data_ptr(), heights.contiguous().data_ptr(),
                                               off_y.contiguous().data_ptr(), off_x.contiguous().data_ptr(), k, int(map_w), int(stride),
                                               float(wh_ratio), int(img_shape[0]), int(img_shape[1]), float(score_thr), dets.data_ptr(),
                                               cnt.data_ptr(), self._stream()), "csp_decode")
            self._check(self.lib.bc_nms_sorted_dev(dets.data_ptr(), k, cnt.data_ptr(), float(iou_thr), ws.data_ptr(), keep.data_ptr(),
                                                   cnt.data_ptr() + 4, self._stream()), "nms_sorted_dev")
        n_keep = int(cnt[1].item())
        return dets[keep[:min(n_keep, int(max_out))].long()]

    def csp_score_monotone_violations(self) -> int:
        """The self-test bc_csp_topk_decode's selection rests on (every neighbouring float pair: the fp32 sigmoid never decreases)."""
        v = torch.zeros(1, dtype=torch.int64, device="cuda")
        self._check(self.lib.bc_csp_score_monotone(v.data_ptr(), self._stream()), "csp_score_monotone")
        return int(v.item())

    def csp_topk_decode_nms(self, cls_map, reg_map, off_map, k, stride, wh_ratio, img_shape, score_thr, iou_thr, max_out, return_top=False):
        """The whole decode from the head's maps to the kept boxes in TWO launches (bc_csp_topk_decode + bc_nms_sorted_dev) and one read of
        the kept count: ``cls_map`` (h, w) centre logits (any supported float type), ``reg_map`` (h, w) float scale predictions, ``off_map``
        (2, h, w) float offsets in any dense layout (strides are passed on).  Equal scores: larger logit first, then lowest position."""
        h, w = cls_map.shape[-2:]
        n = h * w
        assert cls_map.numel() == n and reg_map.numel() == n and off_map.numel() == 2 * n and off_map.shape[0] == 2 and 0 < k <= min(n, 4096)
        assert cls_map.is_cuda and cls_map.dtype in _DTYPE_CODE and cls_map.is_contiguous() and _ok(reg_map, torch.float32) and reg_map.is_contiguous()
        assert off_map.dtype == torch.float32 and off_map.stride(1) == w * off_map.stride(2)
        dev = cls_map.device
        dets = torch.empty((k, 5), dtype=torch.float32, device=dev)
        cnt = torch.empty(2, dtype=torch.int32, device=dev)           # [selected, kept]
        ws = torch.empty(k * ((k + 63) // 64) + k, dtype=torch.int64, device=dev)
        keep = torch.empty(k, dtype=torch.int32, device=dev)
        top = torch.empty(k, dtype=torch.int32, device=dev) if return_top else None
        with torch.cuda.device_of(cls_map):
            self._check(self.lib.bc_csp_topk_decode(cls_map.data_ptr(), _DTYPE_CODE[cls_map.dtype], reg_map.data_ptr(), off_map.data_ptr(),
                                                    off_map.stride(0), off_map.stride(2), n, int(k), int(w), int(stride), float(wh_ratio),
                                                    int(img_shape[0]), int(img_shape[1]), float(score_thr), dets.data_ptr(), cnt.data_ptr(),
                                                    top.data_ptr() if top is not None else None, self._stream()), "csp_topk_decode")
            self._check(self.lib.bc_nms_sorted_dev(dets.data_ptr(), k, cnt.data_ptr(), float(iou_thr), ws.data_ptr(), keep.data_ptr(),
                                                   cnt.data_ptr() + 4, self._stream()), "nms_sorted_dev")
        n_keep = int(cnt[1].item())
        out = dets[keep[:min(n_keep, int(max_out))].long()]
        return (out, top, dets, cnt) if return_top else out

    # -- C. measurement -----------------------------------------------------------------------------------
    def tune(self, key: str, value: int):
        """A/B knob of the library (include/blockcopy_hip.h bc_tune_set): conv_impl, conv2_cfg, conv2_min_lds."""
        if key == "conv2_cfg":
            self._conv_cfg_pinned = self._conv_cfg = int(value)
        self._check(self.lib.bc_tune_set(key.encode(), int(value)), "tune_set")

    def tune_ptr(self, key: str, tensor):
        self._check(self.lib.bc_tune_set_ptr(key.encode(), tensor.data_ptr() if tensor is not None else None), "tune_set_ptr")

    def tune_get(self, key: str) -> int:
        v = ctypes.c_int(0)
        self._check(self.lib.bc_tune_get(key.encode(), ctypes.byref(v)), "tune_get")
        return v.value

    def prof_enable(self, ops=()):
        mask = 0
        for op in ops:
            mask |= 1 << (OP_NAMES.index(op) if isinstance(op, str) else int(op))
        self._check(self.lib.bc_prof_enable(mask), "prof_enable")

    def prof_reset(self):
        self._check(self.lib.bc_prof_reset(), "prof_reset")

    def prof_read(self, op):
        op = OP_NAMES.index(op) if isinstance(op, str) else int(op)
        n, ms, by = ctypes.c_longlong(0), ctypes.c_double(0), ctypes.c_double(0)
        self._check(self.lib.bc_prof_read(op, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)), "prof_read")
        aux = ctypes.c_double(0)
        self._check(self.lib.bc_prof_read_aux(op, ctypes.byref(aux)), "prof_read_aux")
        return dict(op=OP_NAMES[op], launches=n.value, total_ms=ms.value, total_bytes=by.value, total_aux=aux.value)


_backend = None


def get_backend():
    """The active backend; loads the HIP library on first use and fails loudly if that is impossible."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(backend):
    """Install a backend object (test hook; see module docstring).  ``None`` restores lazy HIP loading."""
    global _backend
    prev, _backend = _backend, backend
    return prev
