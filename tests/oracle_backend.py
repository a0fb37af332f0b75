"""Checker backend for CPU tests: the block-op interface of ``blockcopy.backend.HipBackend`` implemented with the
CPU oracle (oracle/oracle.py).  TEST INFRASTRUCTURE ONLY -- lets the host logic (state machine, op routing, index
tables, ring-cache bookkeeping) be exercised against the golden fixtures on a machine without a GPU.  The fused ops
are expressed through the oracle's reference decomposition, which is exactly the equivalence the GPU tests assert
for the HIP kernels."""
from __future__ import annotations

import numpy as np
import torch

import oracle as O


def ring_to_tiles(ring, bs, p):
    """compact ring records (n, C, 4*p*bs) -> (n, C, bs, bs) tiles whose border ring is filled (interior NaN)."""
    n, C, _ = ring.shape
    t = torch.full((n, C, bs, bs), float("nan"), dtype=ring.dtype) if ring.dtype.is_floating_point else torch.zeros((n, C, bs, bs), dtype=ring.dtype)
    T, B, L, R = ring.split([p * bs, p * bs, bs * p, bs * p], dim=2)
    t[:, :, :, :p] = L.reshape(n, C, bs, p)
    t[:, :, :, bs - p:] = R.reshape(n, C, bs, p)
    t[:, :, :p, :] = T.reshape(n, C, p, bs)
    t[:, :, bs - p:, :] = B.reshape(n, C, p, bs)
    return t


def tiles_to_ring(tiles, p):
    n, C, bs, _ = tiles.shape
    return torch.cat([tiles[:, :, :p, :].reshape(n, C, -1), tiles[:, :, bs - p:, :].reshape(n, C, -1),
                      tiles[:, :, :, :p].reshape(n, C, -1), tiles[:, :, :, bs - p:].reshape(n, C, -1)], dim=2)


def _nhwc(x):
    return x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)


def _like(t, ref):
    """t in ref's memory layout (channels-last tensors are checked through their NCHW image)."""
    return t.contiguous(memory_format=torch.channels_last) if _nhwc(ref) else t.contiguous()


def _dynamic(fn):
    """Checker form of a ceiling-sized launch with a device-side executed-tile count (HipBackend._arm / bc_dyn_set): with
    ``dyn = (n_exec_dev, ceiling)`` the op is run on the first ``n = *n_exec_dev`` packed rows -- every 4-D tensor argument with
    ``ceiling`` rows (also inside prologue / epilogue tuples) and the 1-D ``mapping_exec`` are cut to ``n`` -- and the result is padded
    back to ``ceiling`` rows of NaN (rows nobody may consume: a consumer that does becomes loud)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, dyn=None, **kwargs):
        if dyn is None:
            return fn(self, *args, **kwargs)
        n, ceiling = int(dyn[0][0]), int(dyn[1])
        assert 0 <= n <= ceiling

        def cut(a):
            if isinstance(a, torch.Tensor) and ((a.dim() == 4 and a.shape[0] == ceiling) or (a.dim() == 1 and a.dtype == torch.int32 and a.numel() == ceiling)):
                return a[:n]
            if isinstance(a, tuple):
                return tuple(cut(v) for v in a)
            return a

        def grow(r):
            if isinstance(r, torch.Tensor) and r.dim() == 4 and r.shape[0] == n:
                full = torch.full((ceiling,) + tuple(r.shape[1:]), float("nan"), dtype=r.dtype).contiguous(memory_format=torch.channels_last if _nhwc(r) else torch.contiguous_format)
                full[:n] = r
                return full
            if isinstance(r, tuple):
                return tuple(grow(v) for v in r)
            return r

        out = fn(self, *[cut(a) for a in args], **{k: cut(v) for k, v in kwargs.items()})
        if fn.__name__ == "split":
            return args[0]                       # (filled in place through the cut view)
        return grow(out)

    return wrapper


class OracleBackend:
    name = "oracle-cpu"

    @_dynamic
    def split(self, blocks, image, mapping_exec, grid_idx):
        if mapping_exec.numel():
            tmp = torch.empty(blocks.shape, dtype=blocks.dtype)
            O.c_split(tmp, image.contiguous(), mapping_exec)
            blocks.copy_(tmp)
        return blocks

    @_dynamic
    def combine(self, blocks, out, grid_idx, mapping_exec):
        if mapping_exec.numel():
            tmp = out.contiguous()
            O.c_combine(blocks.contiguous(), tmp, mapping_exec)
            out.copy_(tmp)
        return out

    def transfer(self, out, prev_computed, prev_transfer, prev_grid_idx, transfer_idx, padding):
        out.fill_(float("nan"))   # make the don't-care interior loud
        if transfer_idx.numel():
            O.c_transfer(out, prev_computed.contiguous(), prev_transfer.contiguous(), tuple(prev_grid_idx.shape), transfer_idx, padding)
        return out

    def pad(self, data_exec, data_transfer, grid_idx, mapping_exec, pad):
        B, C, bs, _ = data_exec.shape
        out = torch.empty((B, C, bs + 2 * pad, bs + 2 * pad), dtype=data_exec.dtype)
        if mapping_exec.numel():
            O.c_repad(out, data_exec.contiguous(), data_transfer.contiguous(), grid_idx, mapping_exec, pad)
        return out

    def combine_copy(self, blocks, prev, out, grid_idx):
        tmp = prev.contiguous().clone()   # clone + scatter == the reference's non-in-place combine
        mapping = torch.nonzero(grid_idx.reshape(-1) >= 0).squeeze(1).to(torch.int32)
        if mapping.numel():
            O.c_combine(blocks.contiguous(), tmp, mapping)
        out.copy_(tmp)
        return out

    def combine_copy_indirect(self, blocks, slots, grid_idx, out_shape, targets=None):
        """Checker form of the in-graph scatter+copy: the slot words must hold the addresses of ``targets`` (what the HIP kernel
        would dereference); the arithmetic is combine_copy on those tensors."""
        prev, out = targets
        words = slots.tolist()
        assert words[0] == prev.data_ptr() and words[1] == out.data_ptr(), "slot words do not point at this frame's prev / out maps"
        assert tuple(out.shape) == tuple(out_shape)
        if prev is out:      # start of a clip: every tile executed, prev never read
            assert bool((grid_idx >= 0).all())
            prev = torch.zeros_like(out)
        return self.combine_copy(blocks, prev, out, grid_idx)

    def tile_copy_indirect(self, dst, src_slot, mapping_exec, bs, n_exec_dev=None, target=None):
        """Checker form of the in-graph input stage: the slot word must hold the address of ``target`` (what the HIP kernel would
        dereference); the arithmetic is the oracle's split followed by its in-place combine."""
        assert int(src_slot[0]) == target.data_ptr() and target.shape == dst.shape and target.is_contiguous()
        n = mapping_exec.numel() if n_exec_dev is None else min(mapping_exec.numel(), int(n_exec_dev[0]))
        if n:
            m = mapping_exec[:n].contiguous()
            blocks = torch.empty((n, dst.shape[1], bs, bs), dtype=dst.dtype)
            O.c_split(blocks, target, m)
            O.c_combine(blocks, dst, m)
        return dst

    @_dynamic
    def pad_ring(self, data_exec, ring, grid_idx, mapping_exec, pad, prologue=None):
        if _nhwc(data_exec):   # ring records are opaque to the host: the checker keeps its own (NCHW-style) convention
            return _like(self.pad_ring(data_exec.contiguous(), ring, grid_idx, mapping_exec, pad, prologue), data_exec)
        gi = grid_idx.reshape(-1)
        bs = data_exec.shape[2]
        skipped = torch.nonzero(gi < 0).squeeze(1)
        # skipped tiles, compacted in raster order == row (grid_idx + n_total) of the reference's transfer tensor
        transfer = ring_to_tiles(ring[skipped], bs, pad).contiguous()
        # the ring cache keeps what the padded op sees (ACTIVATED values): a prologue transforms the packed tiles only, ring
        # records are never transformed again, and the image-border zeros stay zero because the ACTIVATED tensor is padded
        act = data_exec if prologue is None else self.affine_act(data_exec, prologue[0], prologue[1], None, prologue[2])
        out = self.pad(act, transfer, grid_idx, mapping_exec, pad)
        ring[mapping_exec.long()] = tiles_to_ring(act, pad)
        return out

    supports_fusion_dtypes = (torch.float32,)

    @_dynamic
    def affine_act(self, data, scale=None, shift=None, add=None, relu=False):
        y = data.float()
        if scale is not None:
            y = y * scale.view(1, -1, 1, 1)
        if shift is not None:
            y = y + shift.view(1, -1, 1, 1)
        if add is not None:
            y = y + add.float()
        if relu:
            y = torch.relu(y)
        return _like(y.to(data.dtype), data)

    # ---- by-composition checkers of the fused MI355X forms (the GPU tests assert exactly these equivalences bit for bit)
    @staticmethod
    def pad_ring_add_supported(data_exec, add):
        return _nhwc(data_exec) and add.shape == data_exec.shape and add.dtype == data_exec.dtype

    @_dynamic
    def pad_ring_add(self, data_exec, add, ring, grid_idx, mapping_exec, pad, prologue):
        """residual gather == fused affine pass, then a plain halo gather of its result (ring keeps the activated values)."""
        scale, shift, relu = prologue
        act = self.affine_act(data_exec, scale, shift, add, relu)
        return self.pad_ring(act, ring, grid_idx, mapping_exec, pad, None), act

    @staticmethod
    def maxpool3x3s2_supported(data_exec):
        return _nhwc(data_exec) and data_exec.shape[2] == data_exec.shape[3] and data_exec.shape[2] % 2 == 0

    @_dynamic
    def maxpool3x3s2_ring(self, data_exec, ring, grid_idx, mapping_exec, prologue=None):
        """fused halo + pool == halo gather, then the stock pad-0 pool."""
        padded = self.pad_ring(data_exec, ring, grid_idx, mapping_exec, 1, prologue)
        return _like(torch.nn.functional.max_pool2d(padded.contiguous(), 3, 2, 0), data_exec)

    # ---- fused halo + 3x3 conv: by composition (halo gather, then the stock conv, then the fused epilogue arithmetic)
    @staticmethod
    def conv3x3_supported(data_exec, weight, stride=1, padding=1, dilation=1, groups=1):
        one = lambda v: v if isinstance(v, int) else (v[0] if len(set(v)) == 1 else None)
        return (_nhwc(data_exec) and tuple(weight.shape[2:]) == (3, 3) and one(stride) in (1, 2) and one(padding) == one(dilation) and one(dilation) in (1, 2)
                and (one(dilation) == 1 or one(stride) == 1) and groups == 1 and data_exec.shape[2] % one(stride) == 0 and data_exec.dtype == torch.float32)

    @staticmethod
    def pack_conv3x3_weights(weight):
        return weight.detach().contiguous().reshape(-1)        # the checker keeps the plain (Cout, Cin, 3, 3) order

    @_dynamic
    def conv3x3_ring(self, data_exec, ring, wpk, cout, grid_idx, mapping_exec, prologue=None, epilogue=None, cfg=None, stride=1, dilation=1):
        padded = self.pad_ring(data_exec, ring, grid_idx, mapping_exec, dilation, prologue)
        w = wpk.reshape(cout, data_exec.shape[1], 3, 3)
        y = torch.nn.functional.conv2d(padded.contiguous(), w, stride=stride, dilation=dilation)
        if epilogue is not None:
            scale, shift, add, relu = epilogue
            y = self.affine_act(y, scale, shift, add.contiguous() if add is not None else None, relu)
        return _like(y, data_exec)

    def conv1x1_supported(self, data, weight, stride=1, padding=0, dilation=1, groups=1):
        one = lambda v: v if isinstance(v, int) else (v[0] if len(set(v)) == 1 else None)
        return (data.dim() == 4 and _nhwc(data) and tuple(weight.shape[2:]) == (1, 1) and one(stride) in (1, 2) and one(padding) == 0
                and one(dilation) == 1 and groups == 1 and data.dtype == torch.float32 and (data.shape[0] * data.shape[2] * data.shape[3]) % 64 == 0)

    @_dynamic
    def conv1x1(self, data, wpk, cout, prologue=None, epilogue=None, cfg=None, stride=1):
        x = data
        if prologue is not None:
            x = self.affine_act(x, prologue[0], prologue[1], None, prologue[2])
        y = torch.nn.functional.conv2d(x.contiguous(), wpk.reshape(cout, data.shape[1], 1, 1), stride=stride)
        if epilogue is not None:
            scale, shift, add, relu = epilogue
            y = self.affine_act(y, scale, shift, add.contiguous() if add is not None else None, relu)
        return y.contiguous(memory_format=torch.channels_last)

    # pyramid pooling in two steps (checker form of bc_spp_levels_nhwc / bc_spp_fuse_nhwc: the stock ops, composed)
    @staticmethod
    def spp_supported(x, co, n_levels, cout, grids=None):
        return x.dim() == 4 and x.shape[0] == 1 and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and 1 <= n_levels <= 4 and cout % 64 == 0

    @staticmethod
    def pack_spp_level_weights(weights):
        return torch.stack([w.detach().float().reshape(w.shape[0], w.shape[1]).t() for w in weights]).contiguous()

    @staticmethod
    def pack_spp_fuse_weights(weight):
        return weight.detach().float().clone()

    def spp_levels(self, x, scale, shift, w, grids):
        F = torch.nn.functional
        B = x.shape[0]
        out = []
        for l, (gh, gw) in enumerate(grids):
            p = F.adaptive_avg_pool2d(x.contiguous(), (gh, gw))
            if scale is not None:
                p = p * scale[l].view(1, -1, 1, 1)
            if shift is not None:
                p = p + shift[l].view(1, -1, 1, 1)
            p = torch.relu(p)
            out.append(p.permute(0, 2, 3, 1).reshape(B, gh * gw, -1) @ w[l])
        lv = torch.cat(out, 1).contiguous()
        return lv if B > 1 else lv[0]

    def spp_fuse(self, x, lv, scale, shift, wpk, grids, cout, packed=None):
        if packed is not None:      # checker form of bc_spp_fuse_packed_nhwc: the dense result, then the tiles of the executed positions
            mapping_exec, bs = packed
            dense = type(self).spp_fuse(self, x, lv, scale, shift, wpk, grids, cout)      # (the class's method: an instance-level spy counts calls)
            GW = dense.shape[3] // bs
            tiles = [dense[0, :, (int(ig) // GW) * bs:(int(ig) // GW + 1) * bs, (int(ig) % GW) * bs:(int(ig) % GW + 1) * bs] for ig in mapping_exec.tolist()]
            out = torch.stack(tiles) if tiles else dense.new_zeros((0, cout, bs, bs))
            return out.contiguous(memory_format=torch.channels_last)
        F = torch.nn.functional
        B = x.shape[0]
        H, W = x.shape[2:]
        lv = lv.reshape(B, -1, lv.shape[-1])
        parts, b0 = [x.contiguous()], 0
        for gh, gw in grids:
            m = lv[:, b0:b0 + gh * gw].reshape(B, gh, gw, -1).permute(0, 3, 1, 2).contiguous()
            parts.append(F.interpolate(m, (H, W), mode="bilinear", align_corners=False))
            b0 += gh * gw
        cat = torch.cat(parts, 1)
        if scale is not None:
            cat = cat * scale.view(1, -1, 1, 1)
        if shift is not None:
            cat = cat + shift.view(1, -1, 1, 1)
        return F.conv2d(torch.relu(cat), wpk).contiguous(memory_format=torch.channels_last)

    # dense 3x3 conv to <= 4 channels on a map handed out by to_tensor (checker form of bc_pred3x3_nhwc: the library conv)
    @staticmethod
    def pred3x3_supported(x, weight, stride=1, padding=1, dilation=1, groups=1):
        one = lambda v: v if isinstance(v, (int, str)) else (v[0] if len(set(v)) == 1 else None)
        return (x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last) and tuple(weight.shape[2:]) == (3, 3)
                and one(stride) == 1 and one(padding) == 1 and one(dilation) == 1 and groups == 1 and weight.shape[1] == x.shape[1]
                and x.shape[1] % 32 == 0 and 1 <= weight.shape[0] <= 4)

    @staticmethod
    def pack_pred3x3_weights(weight):
        return weight.detach().float().permute(1, 2, 3, 0).contiguous()

    def pred3x3(self, x, wpk, bias, cout):
        self.calls.append("pred3x3") if hasattr(self, "calls") else None
        w = wpk.permute(3, 0, 1, 2).contiguous()
        return torch.nn.functional.conv2d(x, w, bias, padding=1).contiguous(memory_format=torch.channels_last)

    @staticmethod
    def head1x1_supported(data, weight, stride=1, padding=0, dilation=1, groups=1):
        one = lambda v: v if isinstance(v, int) else (v[0] if len(set(v)) == 1 else None)
        return (data.dim() == 4 and _nhwc(data) and tuple(weight.shape[2:]) == (1, 1) and one(stride) == 1 and one(padding) == 0 and one(dilation) == 1
                and groups == 1 and data.dtype == torch.float32 and weight.shape[0] <= 32 and data.shape[2] == data.shape[3])

    @staticmethod
    def pack_head1x1_weights(weight):
        return weight.detach().clone()

    @_dynamic
    def head1x1(self, data, wpk, cout, prologue=None, epilogue=None, cfg=None, stride=1):
        return self.conv1x1(data, wpk, cout, prologue, epilogue)

    @_dynamic
    def head1x1_scatter(self, data, wpk, cout, prologue, bias, grid_idx, mapping_exec, prev=None, out=None, slots=None, targets=None):
        """Checker form: pointwise conv by composition, then clone + scatter (the reference's non-in-place combine)."""
        if slots is not None:
            prev, out = targets
            words = slots.tolist()
            assert words[0] == prev.data_ptr() and words[1] == out.data_ptr(), "slot words do not point at this frame's prev / out maps"
            if prev is out:
                prev = None
        y = self.conv1x1(data, wpk, cout, prologue, None if bias is None else (None, bias, None, False))
        if prev is None:
            assert bool((grid_idx >= 0).all())
            prev = torch.zeros_like(out)
        self.combine_copy(y, prev, out, grid_idx)
        return out

    @staticmethod
    def group_norm_affine_supported(data, groups):
        return data.dim() == 4 and _nhwc(data) and data.shape[1] % groups == 0 and data.dtype == torch.float32

    def group_norm_affine(self, data, groups, weight=None, bias=None, eps=1e-5):
        """the batched group_norm (statistics over all tiles) as its per-channel affine map, from the definition in fp64"""
        B, C, H, W = data.shape
        x = data.double().permute(1, 0, 2, 3).reshape(groups, -1)
        mean, var = x.mean(1), x.var(1, unbiased=False)
        rstd = (var + eps).rsqrt().repeat_interleave(C // groups)
        mean = mean.repeat_interleave(C // groups)
        g = weight.double() if weight is not None else torch.ones(C, dtype=torch.float64)
        b = bias.double() if bias is not None else torch.zeros(C, dtype=torch.float64)
        scale = g * rstd
        return scale.float(), (b - mean * scale).float()

    supports_interp_dtypes = (torch.float32,)

    @staticmethod
    def interp_epilogue_supported(data):
        return _nhwc(data)

    @_dynamic
    def interp_bilinear(self, data, out_h, out_w, align_corners, rh, rw, epilogue=None):
        # floating-point op: the checker is stock PyTorch on the packed batch (per tile, no halo)
        y = _like(torch.nn.functional.interpolate(data.contiguous(), size=(out_h, out_w), mode="bilinear",
                                                  align_corners=align_corners), data)
        if epilogue is not None:   # deferred interpolation == resample, then the fused affine pass
            scale, shift, add, relu = epilogue
            y = self.affine_act(y, scale, shift, add, relu)
        return y

    def nms(self, dets, iou_thr):
        inds = torch.from_numpy(O.c_nms(dets.detach().float().numpy(), iou_thr))
        return dets[inds, :], inds

    def grid_tables_host(self, grid_u8, grid_idx, mapping, prev_grid_idx=None, transfer=None):
        gi, m = O.c_grid_mappings(grid_u8.astype(bool).reshape(1, 1, 1, -1))
        grid_idx[:] = gi.reshape(-1)
        mapping[:len(m)] = m
        if prev_grid_idx is not None:
            t = O.c_transfer_idx(prev_grid_idx, grid_u8.astype(bool))
            transfer[:len(t)] = t
        return len(m)
