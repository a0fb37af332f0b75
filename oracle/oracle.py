"""CPU ORACLE -- test infrastructure, NOT product code.

Two independent restatements of the reference's block ops:

* ``c_*``  : ctypes bindings to ``bc_oracle.c`` (pixel-index walk, literally the
             kernel strings of blockcopy/blockcopy/utils/block_funcs.py:57-83,
             :130-158, :201-237 and utils/blockpad.py:77-156).
* ``np_*`` : numpy tile-slicing restatements of the same semantics, written from
             the kernels' *meaning*  rather than their index
             arithmetic.  tests/test_oracle.py checks the two against each other
             and against the golden fixtures produced by the reference's Python.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu leg may import
this module; the product package never does (it fails loudly without the HIP
library instead).  Parity pin status: see the header of bc_oracle.c.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbc_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile bc_oracle.c with gcc (idempotent)."""
    src = os.path.join(_HERE, "bc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libbc_oracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        i, p = ctypes.c_int, ctypes.c_void_p
        _lib.bco_split.argtypes = [p, p, p] + [i] * 7
        _lib.bco_combine.argtypes = [p, p, p] + [i] * 7
        _lib.bco_transfer.argtypes = [p, p, p, p] + [i] * 8
        _lib.bco_repad.argtypes = [p, p, p, p, p] + [i] * 8
        _lib.bco_grid_mappings.argtypes = [p, i, p, p]
        _lib.bco_grid_mappings.restype = i
        _lib.bco_transfer_idx.argtypes = [p, p, i, p]
        _lib.bco_transfer_idx.restype = i
        _lib.bco_nms_sorted.argtypes = [p, i, ctypes.c_float, p]
        _lib.bco_nms_sorted.restype = i
        for f in (_lib.bco_split, _lib.bco_combine, _lib.bco_transfer, _lib.bco_repad):
            f.restype = None
    return _lib


def _np(x) -> np.ndarray:
    """numpy view of a numpy array or a CPU torch tensor (shares memory)."""
    if isinstance(x, np.ndarray):
        return x
    x = x.detach()
    if str(x.dtype) == "torch.bfloat16":   # numpy has no bf16; the block ops are pure copies => raw 16-bit words
        import torch

        x = x.view(torch.int16)
    return x.numpy()


def _ptr(a: np.ndarray) -> ctypes.c_void_p:
    assert a.flags["C_CONTIGUOUS"], "oracle expects contiguous arrays (reference asserts is_contiguous, utils/cuda.py:46)"
    return ctypes.c_void_p(a.ctypes.data)


def _i32(a) -> np.ndarray:
    a = _np(a)
    assert a.dtype == np.int32, a.dtype
    return np.ascontiguousarray(a)


# --------------------------------------------------------------------------- C bindings
def c_split(blocks, image, mapping_exec, grid_shape=None):
    """reference: SplitFunction.forward, utils/block_funcs.py:10-49.  Fills and returns ``blocks``."""
    b, im, m = _np(blocks), _np(image), _i32(mapping_exec)
    N, C, H, W = im.shape
    BS = b.shape[2]
    assert b.shape[1] == C and b.shape[2] == b.shape[3] and b.dtype == im.dtype
    lib().bco_split(_ptr(b), _ptr(im), _ptr(m), len(m), N, C, H, W, BS, b.itemsize)
    return blocks


def c_combine(blocks, out, mapping_exec):
    """reference: CombineFunction.forward, utils/block_funcs.py:85-124.  Mutates and returns ``out``."""
    b, o, m = _np(blocks), _np(out), _i32(mapping_exec)
    N, C, H, W = o.shape
    BS = b.shape[2]
    assert b.dtype == o.dtype
    lib().bco_combine(_ptr(b), _ptr(o), _ptr(m), len(m), N, C, H, W, BS, b.itemsize)
    return out


def c_transfer(out, prev_computed, prev_transfer, prev_grid_idx_shape, transfer_idx, padding):
    """reference: TransferFunction.forward, utils/block_funcs.py:161-193.  ``prev_grid_idx_shape`` = (N,1,GH,GW)."""
    o, pc, pt, t = _np(out), _np(prev_computed), _np(prev_transfer), _i32(transfer_idx)
    N, _, GH, GW = prev_grid_idx_shape
    _, C, BS, _ = o.shape
    assert o.dtype == pc.dtype == pt.dtype
    lib().bco_transfer(_ptr(o), _ptr(pc), _ptr(pt), _ptr(t), len(t), N, C, GH, GW, BS, int(padding), o.itemsize)
    return out


def c_repad(out, features, transfer, grid_idx, mapping_exec, pad):
    """reference: BlockPadFunction.forward, utils/blockpad.py:21-71 (``out`` pre-allocated by the caller here)."""
    o, f, t, g, m = _np(out), _np(features), _np(transfer), _i32(grid_idx), _i32(mapping_exec)
    N, _, GH, GW = g.shape
    _, C, BS, _ = f.shape
    assert o.shape == (f.shape[0], C, BS + 2 * pad, BS + 2 * pad), (o.shape, f.shape, pad)
    assert o.dtype == f.dtype == t.dtype
    lib().bco_repad(_ptr(o), _ptr(f), _ptr(t), _ptr(g), _ptr(m), len(m), N, C, GH, GW, BS, int(pad), o.itemsize)
    return out


def c_grid_mappings(grid):
    """reference: get_grid_mappings, core/tensorwrapper.py:108-128.  grid: bool (N,1,GH,GW)."""
    g = np.ascontiguousarray(_np(grid).astype(np.uint8))
    n_total = g.size
    grid_idx = np.empty(g.shape, np.int32)
    mapping = np.empty(n_total, np.int32)
    n_exec = lib().bco_grid_mappings(_ptr(g), n_total, _ptr(grid_idx), _ptr(mapping))
    return grid_idx, mapping[:n_exec].copy()


def c_transfer_idx(prev_grid_idx, grid):
    """reference: core/tensorwrapper.py:176-178."""
    g = np.ascontiguousarray(_np(grid).astype(np.uint8))
    p = _i32(prev_grid_idx)
    out = np.empty(g.size, np.int32)
    n = lib().bco_transfer_idx(_ptr(p), _ptr(g), g.size, _ptr(out))
    return out[:n].copy()


def c_policy_step(logits, seed, counter, multiple, at_least_one=False):
    """CPU restatement of the product's device policy step (include/blockcopy_hip.h bc_policy_step): returns
    (grid bool like logits, counts int32[3] = {n_exec, n_sampled, nan flag}, probs float32)."""
    import ctypes

    x = np.ascontiguousarray(_np(logits), dtype=np.float32)
    grid = np.empty(x.size, np.uint8)
    counts = np.zeros(3, np.int32)
    probs = np.empty(x.size, np.float32)
    fn = lib().bco_policy_step
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    fn(_ptr(x), x.size, int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1), int(multiple), int(bool(at_least_one)), _ptr(grid),
       _ptr(counts), _ptr(probs))
    return grid.astype(bool).reshape(x.shape), counts, probs.reshape(x.shape)


def c_nms(dets, iou_thr):
    """reference: nms_cuda, Pedestron/mmdet/ops/nms/src/nms_kernel.cu:70-130 -- original indices of the kept boxes, ascending.
    dets: (n,5) float32 [x1,y1,x2,y2,score]."""
    d = np.ascontiguousarray(_np(dets), dtype=np.float32)
    n = d.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    order = np.argsort(-d[:, 4], kind="stable")
    srt = np.ascontiguousarray(d[order])
    keep = np.empty(n, np.int32)
    k = lib().bco_nms_sorted(_ptr(srt), n, float(iou_thr), _ptr(keep))
    return np.sort(order[keep[:k]]).astype(np.int64)


# --------------------------------------------------------------------------- numpy restatements
def _tile(i_g, GH, GW):
    return i_g // (GH * GW), (i_g // GW) % GH, i_g % GW


def np_grid_mappings(grid):
    g = _np(grid).astype(bool)
    flat = g.reshape(-1)
    n_total = flat.size
    grid_idx = np.empty(n_total, np.int32)
    grid_idx[flat] = np.arange(flat.sum(), dtype=np.int32)
    grid_idx[~flat] = np.arange(-n_total, -n_total + (~flat).sum(), dtype=np.int32)
    return grid_idx.reshape(g.shape), np.nonzero(flat)[0].astype(np.int32)


def np_split(image, mapping_exec, BS):
    im = _np(image)
    N, C, H, W = im.shape
    GH, GW = H // BS, W // BS
    out = np.empty((len(mapping_exec), C, BS, BS), im.dtype)
    for b, i_g in enumerate(_np(mapping_exec).tolist()):
        n, gh, gw = _tile(i_g, GH, GW)
        out[b] = im[n, :, gh * BS:(gh + 1) * BS, gw * BS:(gw + 1) * BS]
    return out


def np_combine(blocks, out, mapping_exec):
    b, o = _np(blocks), _np(out)
    N, C, H, W = o.shape
    BS = b.shape[2]
    GH, GW = H // BS, W // BS
    for k, i_g in enumerate(_np(mapping_exec).tolist()):
        n, gh, gw = _tile(i_g, GH, GW)
        o[n, :, gh * BS:(gh + 1) * BS, gw * BS:(gw + 1) * BS] = b[k]
    return out


def ring_mask(BS, p):
    """True where transfer_kernel writes (border ring of width p); the interior is a don't-care."""
    m = np.ones((BS, BS), bool)
    if p >= 0 and BS - 2 * p > 0:
        m[p:BS - p, p:BS - p] = False
    return m


def np_transfer(out, prev_computed, prev_transfer, n_total_prev, transfer_idx, padding):
    o, pc, pt = _np(out), _np(prev_computed), _np(prev_transfer)
    BS = o.shape[2]
    m = ring_mask(BS, padding)
    for b, b_prev in enumerate(_np(transfer_idx).tolist()):
        src = pc[b_prev] if b_prev >= 0 else pt[b_prev + n_total_prev]
        o[b][:, m] = src[:, m]
    return out


def np_repad(features, transfer, grid_idx, mapping_exec, p):
    f, t, g = _np(features), _np(transfer), _np(grid_idx)
    N, _, GH, GW = g.shape
    n_total = g.size
    n_exec, C, BS, _ = f.shape
    out = np.zeros((n_exec, C, BS + 2 * p, BS + 2 * p), f.dtype)

    def tile_data(n, gh, gw):
        k = int(g[n, 0, gh, gw])
        return f[k] if k >= 0 else t[k + n_total]

    for b, i_g in enumerate(_np(mapping_exec).tolist()):
        n, gh, gw = _tile(i_g, GH, GW)
        out[b, :, p:p + BS, p:p + BS] = f[b]
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dy == 0 and dx == 0:
                    continue
                nh, nw = gh + dy, gw + dx
                if not (0 <= nh < GH and 0 <= nw < GW):
                    continue  # image border: stays zero
                src = tile_data(n, nh, nw)
                ys = {-1: (slice(0, p), slice(BS - p, BS)), 0: (slice(p, p + BS), slice(0, BS)), 1: (slice(p + BS, 2 * p + BS), slice(0, p))}[dy]
                xs = {-1: (slice(0, p), slice(BS - p, BS)), 0: (slice(p, p + BS), slice(0, BS)), 1: (slice(p + BS, 2 * p + BS), slice(0, p))}[dx]
                out[b, :, ys[0], xs[0]] = src[:, ys[1], xs[1]]
    return out
