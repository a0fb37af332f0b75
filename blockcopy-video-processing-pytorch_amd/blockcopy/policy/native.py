"""The RL policy's CNN on own kernels: forward, REINFORCE backward and RMSprop step without PyTorch in the loop.

Reference: ``PolicyNet`` (policy/net.py:17-125), its trunk (policy/resnet.py:60-115), ``PolicyTrainRL.forward`` / ``.optim``
(policy/policy.py:265-370) and ``InformationGainSemSeg`` (policy/information_gain.py:22-41).  The reference runs the policy every
frame through autograd (batch-statistics BatchNorms, ~45 library launches) and every ``train_interval``-th frame through a backward
pass + optimizer (~120 more), all enqueued from Python.  Here the same arithmetic is a FIXED sequence of csrc/policy_net.hip launches
over preallocated channels-last fp32 buffers -- a forward sequence and a step sequence -- each captured once in a hipGraph:

    per frame      1 gather launch (policy input, channels-last, padded to 32 channels) + 1 graph replay  ->  tile logits
    per train step 1 information-gain launch + 1 graph replay (seed, backward, RMSprop, parameter export)

Parameters live in ONE flat fp32 buffer in the kernels' layout (conv weights [tap][Cin][Cout]); after every step a single launch
exports them into the module's ``nn.Parameter`` tensors (any strides), so ``state_dict()`` / checkpoints / the eager path always see
current values; external writes to the parameters (``load_state_dict``) are detected through their version counters and imported.
The RMSprop state (square average, momentum buffer) lives beside the flat parameters (``state_dict()`` of THIS object), not in the
``torch.optim.RMSprop`` instance, whose hyper-parameters remain the ones used.

Batch statistics follow ``F.batch_norm(training=True)``: biased variance for the normalisation, unbiased for the running estimate,
momentum and eps of the module, ``num_batches_tracked`` incremented on the device.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

ENABLED = os.environ.get("BLOCKCOPY_NATIVE_POLICY", "1") != "0"
USE_GRAPH = os.environ.get("BLOCKCOPY_NATIVE_POLICY_GRAPH", "1") != "0"
# BatchNorm statistics through fixed-point accumulators (no finish launch per layer: bc_pn_arm_bn); 0 = per-workgroup partial sums + bc_pn_bn_finalize
BN_ACC = os.environ.get("BLOCKCOPY_NATIVE_POLICY_BN_ACC", "1") != "0"
CPAD = 32
# forward convs: 1 = split-fp16 products on the 16-bit matrix pipe (fp32-level accuracy, 5.3 x the fp32 matrix rate; bc_pn_conv_nhwc), 0 = fp32 pipe
FWD_PRECISION = int(os.environ.get("BLOCKCOPY_POLICY_FWD_PRECISION", "1"))
# data gradients: 2 = split-bf16 products (no scaling needed over the gradients' range; ~2e-5 relative, against the 1e-3 the REINFORCE step is
# tested to and an RMSprop update that is sign-like), 0 = fp32 pipe
BWD_PRECISION = int(os.environ.get("BLOCKCOPY_POLICY_BWD_PRECISION", "2"))


def _pad32(c: int) -> int:
    return (c + CPAD - 1) // CPAD * CPAD


class _Conv:
    def __init__(self, name: str, mod: nn.Conv2d, N: int, Hx: int, Wx: int):
        self.name, self.mod = name, mod
        self.ks, self.stride = mod.kernel_size[0], mod.stride[0]
        self.Cx, self.Cy = mod.in_channels, mod.out_channels
        self.Cxp = _pad32(self.Cx)
        pad = 1 if self.ks == 3 else 0
        self.N, self.Hx, self.Wx = N, Hx, Wx
        self.Hy, self.Wy = (Hx + 2 * pad - self.ks) // self.stride + 1, (Wx + 2 * pad - self.ks) // self.stride + 1
        self.off = self.off_t = -1       # float offsets into the flat / transposed buffers
        self.z = None                    # raw conv output (N, Hy, Wy, Cy)

    @property
    def numel(self):
        return self.ks * self.ks * self.Cxp * self.Cy


class _BN:
    def __init__(self, name: str, mod: nn.BatchNorm2d, dev):
        self.name, self.mod, self.C = name, mod, mod.num_features
        self.off_g = self.off_b = -1
        v = lambda: torch.zeros(self.C, dtype=torch.float32, device=dev)
        self.scale, self.shift, self.mean, self.invstd = v(), v(), v(), v()
        # fixed-point accumulators [sum x | sum x^2] of the layer's input (bc_pn_arm_bn) and the pixel count of the conv that feeds it
        self.acc = torch.zeros(16 * 2 * self.C, dtype=torch.int64, device=dev)      # (16 replicas: include/blockcopy_hip.h bc_pn_arm_bn)
        self.count = 0.0


class _Block:
    """BasicBlock bookkeeping: input (tensor + optional BN prologue), conv1/bn1, conv2/bn2, optional projection shortcut, output."""

    def __init__(self):
        self.x = self.x_bn = None
        self.c1 = self.b1 = self.c2 = self.b2 = self.cd = self.bd = None
        self.out = None


def supported(net, optimizer) -> Optional[str]:
    """None when the module / optimizer pair is the reference's policy net in a form the kernels cover; else the reason."""
    from blockcopy.policy.resnet import BasicBlock, ResNet_32x32

    bb = getattr(net, "backbone", None)
    if not isinstance(bb, ResNet_32x32):
        return "trunk is not the reference's resnet8"
    for i in (1, 2, 3):
        stage = getattr(bb, f"layer{i}")
        if len(stage) != 1 or not isinstance(stage[0], BasicBlock):
            return "one BasicBlock per stage expected"
        blk = stage[0]
        if blk.conv1.out_channels % 32 or blk.conv1.in_channels % 32 and i > 1:
            return "stage widths must be multiples of 32"
        if (blk.downsample is None) != (blk.stride == 1 and blk.conv1.in_channels == blk.conv1.out_channels):
            return "unexpected shortcut"
    if bb.conv1.out_channels % 32 or bb.conv1.in_channels > CPAD:
        return "stem width"
    if len(net.layers) != 3 or net.layers[2][0].out_channels != 1 or net.layers[2][0].bias is None:
        return "head is not three stride-2 stages ending in one logit"
    for st in net.layers:
        c = st[0]
        if c.kernel_size != (3, 3) or c.stride != (2, 2) or c.padding != (1, 1):
            return "head stage geometry"
    if net.layers[0][0].out_channels % 32 or net.layers[1][0].out_channels % 32:
        return "head widths"
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d) and not (m.affine and m.track_running_stats and m.momentum is not None):
            return "BatchNorm flavour"
        if isinstance(m, nn.Conv2d) and (m.groups != 1 or m.dilation != (1, 1)):
            return "conv flavour"
    if any(p.dtype != torch.float32 for p in net.parameters()):
        return "parameters are not fp32"
    if not isinstance(optimizer, torch.optim.RMSprop) or len(optimizer.param_groups) != 1 or optimizer.param_groups[0].get("centered", False):
        return "optimizer is not a single-group, uncentred RMSprop"
    ids = {id(p) for p in optimizer.param_groups[0]["params"]}
    if ids != {id(p) for p in net.parameters()}:
        return "optimizer does not cover exactly the net's parameters"
    return None


class NativePolicyNet:
    def __init__(self, net, optimizer, frame_shape, device):
        from blockcopy.backend import get_backend

        be = get_backend()
        if be.name != "hip":
            raise RuntimeError("NativePolicyNet needs the HIP backend")
        self.be, self.lib = be, be.lib
        self.net, self.optimizer, self.dev = net, optimizer, torch.device(device)
        N, _, H, W = frame_shape
        self.frame_shape = tuple(frame_shape)
        self.N = N
        self.h, self.w = int(np.floor(H * net.scale_factor)), int(np.floor(W * net.scale_factor))
        self.GH, self.GW = H // net.block_size, W // net.block_size
        self.n_total = N * self.GH * self.GW
        self._plist = self._blist = None      # cached parameter / buffer object lists (the module's structure is fixed once validated)
        self._feat_desc = {}
        self._build_topology()
        self._alloc()
        self._segs_key = None
        self._import_params()
        self._fwd_ops = self._build_forward_acc() if BN_ACC else self._build_forward()
        self._step_ops = self._build_step()
        self._fwd_graph = self._step_graph = None
        self._fwd_warm = self._step_warm = 0
        self._hyper = None
        self.forwards = self.steps = 0

    # ------------------------------------------------------------------------------------------------------------ structure
    def _build_topology(self):
        net, N, dev = self.net, self.N, self.dev
        bb = net.backbone
        self.convs: List[_Conv] = []
        self.bns: List[_BN] = []

        def conv(name, mod, Hx, Wx):
            c = _Conv(name, mod, N, Hx, Wx)
            self.convs.append(c)
            return c

        def bn(name, mod):
            b = _BN(name, mod, dev)
            self.bns.append(b)
            return b

        h, w = self.h, self.w
        self.stem = conv("backbone.conv1", bb.conv1, h, w)
        self.stem_bn = bn("backbone.bn1", bb.bn1)
        self.blocks: List[_Block] = []
        prev_c, prev_bn, prev_t = self.stem, self.stem_bn, None       # block input: (conv whose raw output it is, its BN) or a materialised tensor
        for i in (1, 2, 3):
            m = getattr(bb, f"layer{i}")[0]
            b = _Block()
            b.x_conv, b.x_bn, b.x_t = prev_c, prev_bn, prev_t
            b.c1 = conv(f"backbone.layer{i}.0.conv1", m.conv1, h, w)
            b.b1 = bn(f"backbone.layer{i}.0.bn1", m.bn1)
            h, w = b.c1.Hy, b.c1.Wy
            b.c2 = conv(f"backbone.layer{i}.0.conv2", m.conv2, h, w)
            b.b2 = bn(f"backbone.layer{i}.0.bn2", m.bn2)
            if m.downsample is not None:
                b.cd = conv(f"backbone.layer{i}.0.downsample.0", m.downsample[0], b.c1.Hx, b.c1.Wx)
                b.bd = bn(f"backbone.layer{i}.0.downsample.1", m.downsample[1])
                assert (b.cd.Hy, b.cd.Wy) == (h, w)
            self.blocks.append(b)
            prev_c, prev_bn, prev_t = None, None, b
        self.head0 = conv("layers.0.0", net.layers[0][0], h, w)
        self.head0_bn = bn("layers.0.1", net.layers[0][1])
        self.head1 = conv("layers.1.0", net.layers[1][0], self.head0.Hy, self.head0.Wy)
        self.head1_bn = bn("layers.1.1", net.layers[1][1])
        self.last = net.layers[2][0]
        self.last_Hi, self.last_Wi, self.last_C = self.head1.Hy, self.head1.Wy, self.head1.Cy
        Ho, Wo = (self.last_Hi - 1) // 2 + 1, (self.last_Wi - 1) // 2 + 1
        if (Ho, Wo) != (self.GH, self.GW):
            raise RuntimeError(f"policy logits {(Ho, Wo)} do not tile the frame grid {(self.GH, self.GW)}")

    def _alloc(self):
        dev, N, lib = self.dev, self.N, self.lib
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        self.feat = f(N, self.h, self.w, CPAD)
        max_act = N * self.h * self.w * CPAD
        stats_cap = wg_cap = bwd_cap = 0
        for c in self.convs:
            c.z = f(N, c.Hy, c.Wy, c.Cy)
            max_act = max(max_act, c.z.numel())
            stats_cap = max(stats_cap, lib.bc_pn_conv_partials(N, c.Hy, c.Wy, c.Cy) * 2 * c.Cy)
            c.groups = lib.bc_pn_wgrad_groups(N, c.Hy, c.Wy, c.Cxp, c.Cy)
            c.ws_off, wg_cap = wg_cap, wg_cap + (c.groups * c.numel + 3) // 4 * 4      # every conv keeps its partial copies until the step's last launch sums them
            bwd_cap = max(bwd_cap, lib.bc_pn_bn_bwd_partials(N * c.Hy * c.Wy) * 2 * c.Cy)
        for b in self.blocks:
            b.out = f(N, b.c2.Hy, b.c2.Wy, b.c2.Cy)
        self.stats = f(stats_cap)
        self.wg_ws = f(wg_cap)
        self.bwd_part = f(bwd_cap)
        self.bwd_coef = f(3 * 1024)
        self.gbuf = [f(max_act) for _ in range(3)]
        # flat parameter space: conv weights [tap][Cin_pad][Cout] (16-byte aligned segments), BN gamma / beta, last conv [9][C] + bias
        off = off_t = 0
        al = lambda n: (n + 3) // 4 * 4
        for c in self.convs:
            c.off, c.off_t = off, off_t
            off += al(c.numel)
            off_t += al(c.numel)
        for b in self.bns:
            b.off_g, off = off, off + al(b.C)
            b.off_b, off = off, off + al(b.C)
        self.off_last_w, off = off, off + al(9 * self.last_C)
        self.off_last_b, off = off, off + 4
        self.n_flat = off
        self.P, self.G, self.SQ, self.MOM = f(off), f(off), f(off), f(off)
        self.WT = f(max(off_t, 4))
        self.logits = f(self.n_total)
        self.gl, self.loss, self.reward = f(self.n_total), f(1), f(self.n_total)
        self.grid_u8 = torch.zeros(self.n_total, dtype=torch.uint8, device=dev)
        self.cost = torch.zeros(1, dtype=torch.float64, device=dev)
        self.ig = None          # allocated at the first step (its size follows the task output)
        self._hq = self._wq = 0

    # ------------------------------------------------------------------------------------------------------------ parameters
    def _param_list(self):
        ps = [c.mod.weight for c in self.convs]
        for b in self.bns:
            ps += [b.mod.weight, b.mod.bias]
        ps += [self.last.weight, self.last.bias]
        return ps

    def _buffer_list(self):
        out = []
        for b in self.bns:
            out += [b.mod.running_mean, b.mod.running_var, b.mod.num_batches_tracked]
        return out

    def _build_segs(self):
        rec = np.zeros(0, dtype=np.uint8)
        assert self.lib.bc_pn_seg_bytes() == 96
        dt = np.dtype([("off", "<i8"), ("off_t", "<i8"), ("ptr", "<u8"), ("s_co", "<i8"), ("s_ci", "<i8"), ("s_ky", "<i8"), ("s_kx", "<i8"),
                       ("taps", "<i4"), ("kw", "<i4"), ("cin", "<i4"), ("cin_pad", "<i4"), ("cout", "<i4"), ("numel", "<i4"),
                       ("ws_off", "<i8"), ("groups", "<i4"), ("pad", "<i4")])
        rows = []
        for c in self.convs:
            w = c.mod.weight
            rows.append((c.off, c.off_t, w.data_ptr(), w.stride(0), w.stride(1), w.stride(2), w.stride(3), c.ks * c.ks, c.ks, c.Cx, c.Cxp, c.Cy, c.numel,
                         c.ws_off, c.groups, 0))
        for b in self.bns:
            for off, p in ((b.off_g, b.mod.weight), (b.off_b, b.mod.bias)):
                assert p.is_contiguous()
                rows.append((off, -1, p.data_ptr(), 0, 0, 0, 0, 0, 1, 0, 0, 0, b.C, 0, 0, 0))
        w = self.last.weight
        rows.append((self.off_last_w, -1, w.data_ptr(), w.stride(0), w.stride(1), w.stride(2), w.stride(3), 9, 3, self.last_C, self.last_C, 1, 9 * self.last_C, 0, 0, 0))
        rows.append((self.off_last_b, -1, self.last.bias.data_ptr(), 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0))
        rec = np.array(rows, dtype=dt)
        self.n_segs = len(rows)
        self.segs = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(self.dev)
        self._segs_key = self._ptr_key()

    def _ptr_key(self):
        if self._plist is None:
            self._plist, self._blist = self._param_list(), self._buffer_list()
        return tuple([p.data_ptr() for p in self._plist] + [b.data_ptr() for b in self._blist])

    def _versions(self):
        if self._plist is None:
            self._plist, self._blist = self._param_list(), self._buffer_list()
        return tuple([p._version for p in self._plist])

    def _check(self, rc, what):
        if rc != 0:
            self.be._check(rc, what)

    def _import_params(self):
        """module parameters -> flat buffer (+ transposed copies)."""
        self._build_segs()
        with torch.cuda.device(self.dev):
            self._check(self.lib.bc_pn_sync_params(self.P.data_ptr(), self.WT.data_ptr(), self.segs.data_ptr(), self.n_segs, 1, self._stream()), "pn_sync_params")
        self._seen_versions = self._versions()

    def params_current(self) -> bool:
        return self._segs_key == self._ptr_key() and self._seen_versions == self._versions()

    def _refresh_if_needed(self):
        if self._segs_key != self._ptr_key():
            # the module's tensors moved (``.to()``): every captured pointer is stale
            self._import_params()
            self._fwd_ops, self._step_ops = (self._build_forward_acc() if BN_ACC else self._build_forward()), self._build_step()
            self._fwd_graph = self._step_graph = None
        elif self._seen_versions != self._versions():
            self._import_params()      # somebody wrote the parameters (load_state_dict, an eager optimizer step)

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Optimizer state of the native step (flat, kernel layout)."""
        return {"square_avg": self.SQ.clone(), "momentum_buffer": self.MOM.clone(), "steps": torch.tensor(self.steps)}

    def load_state_dict(self, sd):
        self.SQ.copy_(sd["square_avg"])
        self.MOM.copy_(sd["momentum_buffer"])
        self.steps = int(sd["steps"])

    # ------------------------------------------------------------------------------------------------------------ op builders
    def _conv_fwd(self, c: _Conv, x, pro: Optional[_BN], bn: _BN):
        """conv (per-workgroup partial sums of its output in the epilogue) + the training-mode BatchNorm finish.  (The finish inside the conv
        launch -- last workgroup by ticket -- was built and measured SLOWER: 46-51 us against 41-45 per full-resolution layer, DESIGN.md.)"""
        lib, P = self.lib, self.P
        sc, sh = (pro.scale.data_ptr(), pro.shift.data_ptr()) if pro is not None else (None, None)
        relu = 1 if pro is not None else 0
        n_part = lib.bc_pn_conv_partials(c.N, c.Hy, c.Wy, c.Cy)
        args = (c.z.data_ptr(), x.data_ptr(), P.data_ptr() + 4 * c.off, c.N, c.Hx, c.Wx, c.Cxp, c.Hy, c.Wy, c.Cy, c.ks, c.stride, 0, sc, sh, relu, None, None, 0,
                self.stats.data_ptr(), self.stats.numel(), FWD_PRECISION)
        m = bn.mod
        fin = (self.stats.data_ptr(), n_part, bn.C, float(c.N * c.Hy * c.Wy), P.data_ptr() + 4 * bn.off_g, P.data_ptr() + 4 * bn.off_b, float(m.eps), float(m.momentum),
               m.running_mean.data_ptr(), m.running_var.data_ptr(), m.num_batches_tracked.data_ptr(), bn.scale.data_ptr(), bn.shift.data_ptr(),
               bn.mean.data_ptr(), bn.invstd.data_ptr())

        def run(st):
            self._check(lib.bc_pn_conv_nhwc(*args, st), "pn_conv_nhwc")
            self._check(lib.bc_pn_bn_finalize(*fin, st), "pn_bn_finalize")
        return run

    def _conv_fwd_acc(self, c: _Conv, x, pro: Optional[_BN], bn: _BN):
        """conv whose output statistics go to ``bn``'s fixed-point accumulators and whose prologue derives the coefficients of ``pro`` from ITS
        accumulators: no BatchNorm finish between the layers (bc_pn_arm_bn; one bc_pn_bn_finalize_acc at the end of the pass)."""
        lib, P = self.lib, self.P
        bn.count = float(c.N * c.Hy * c.Wy)
        arm = ((pro.acc.data_ptr(), P.data_ptr() + 4 * pro.off_g, P.data_ptr() + 4 * pro.off_b, pro.count, float(pro.mod.eps), pro.C) if pro is not None
               else (None, None, None, 0.0, 0.0, 0)) + (bn.acc.data_ptr(),)
        assert pro is None or (pro.count > 0 and pro.C == c.Cxp)
        args = (c.z.data_ptr(), x.data_ptr(), P.data_ptr() + 4 * c.off, c.N, c.Hx, c.Wx, c.Cxp, c.Hy, c.Wy, c.Cy, c.ks, c.stride, 0, None, None,
                1 if pro is not None else 0, None, None, 0, None, 0, FWD_PRECISION)

        def run(st):
            self._check(lib.bc_pn_arm_bn(*arm), "pn_arm_bn")
            self._check(lib.bc_pn_conv_nhwc(*args, st), "pn_conv_nhwc")
        return run

    def _build_forward_acc(self):
        """The forward pass without BatchNorm finish launches (11 fewer): see _conv_fwd_acc."""
        lib, P = self.lib, self.P
        gb = lambda bn: (bn.acc.data_ptr(), P.data_ptr() + 4 * bn.off_g, P.data_ptr() + 4 * bn.off_b)
        ops = [self._conv_fwd_acc(self.stem, self.feat, None, self.stem_bn)]
        for b in self.blocks:
            x, pro = self._block_input(b)
            ops.append(self._conv_fwd_acc(b.c1, x, pro, b.b1))
            ops.append(self._conv_fwd_acc(b.c2, b.c1.z, b.b1, b.b2))
            pixels = b.c2.N * b.c2.Hy * b.c2.Wy
            eps = float(b.b2.mod.eps)
            if b.cd is not None:
                ops.append(self._conv_fwd_acc(b.cd, x, pro, b.bd))
                assert b.bd.count == b.b2.count and float(b.bd.mod.eps) == eps
                jargs = (b.out.data_ptr(), b.c2.z.data_ptr()) + gb(b.b2) + (b.cd.z.data_ptr(),) + gb(b.bd) + (b.b2.count, eps, 1, b.c2.Cy, pixels)
            elif pro is not None:
                assert pro.count == b.b2.count and float(pro.mod.eps) == eps
                jargs = (b.out.data_ptr(), b.c2.z.data_ptr()) + gb(b.b2) + (x.data_ptr(),) + gb(pro) + (b.b2.count, eps, 2, b.c2.Cy, pixels)
            else:
                jargs = (b.out.data_ptr(), b.c2.z.data_ptr()) + gb(b.b2) + (x.data_ptr(), None, None, None, b.b2.count, eps, 0, b.c2.Cy, pixels)
            ops.append(lambda st, a=jargs: self._check(lib.bc_pn_join_acc(*a, st), "pn_join_acc"))
        ops.append(self._conv_fwd_acc(self.head0, self.blocks[-1].out, None, self.head0_bn))
        ops.append(self._conv_fwd_acc(self.head1, self.head0.z, self.head0_bn, self.head1_bn))
        h1 = self.head1_bn
        hargs = (self.logits.data_ptr(), self.head1.z.data_ptr()) + gb(h1) + (h1.count, float(h1.mod.eps), P.data_ptr() + 4 * self.off_last_w,
                                                                               P.data_ptr() + 4 * self.off_last_b, self.N, self.last_Hi, self.last_Wi, self.last_C)
        ops.append(lambda st: self._check(lib.bc_pn_head_fwd_acc(*hargs, st), "pn_head_fwd_acc"))
        # end of the pass: the arrays the backward pass reads, running statistics, batch counters; accumulators zeroed
        rec = np.dtype([("acc", "<i8"), ("gamma", "<i8"), ("beta", "<i8"), ("rm", "<i8"), ("rv", "<i8"), ("batches", "<i8"), ("scale", "<i8"), ("shift", "<i8"),
                        ("mean", "<i8"), ("invstd", "<i8"), ("count", "<f8"), ("eps", "<f4"), ("momentum", "<f4"), ("C", "<i4"), ("pad", "<i4")])
        assert rec.itemsize == lib.bc_pn_bn_layer_bytes()
        tab = np.zeros(len(self.bns), dtype=rec)
        for k, bn in enumerate(self.bns):
            assert bn.count > 0, bn.name
            m = bn.mod
            tab[k] = (bn.acc.data_ptr(), P.data_ptr() + 4 * bn.off_g, P.data_ptr() + 4 * bn.off_b, m.running_mean.data_ptr(), m.running_var.data_ptr(),
                      m.num_batches_tracked.data_ptr(), bn.scale.data_ptr(), bn.shift.data_ptr(), bn.mean.data_ptr(), bn.invstd.data_ptr(), bn.count,
                      float(m.eps), float(m.momentum), bn.C, 0)
        self._bn_layers = torch.from_numpy(tab.view(np.uint8).copy()).to(self.P.device)
        n_layers, layers_ptr = len(self.bns), self._bn_layers.data_ptr()
        ops.append(lambda st: self._check(lib.bc_pn_bn_finalize_acc(layers_ptr, n_layers, st), "pn_bn_finalize_acc"))
        return ops

    def _block_input(self, b: _Block):
        """(tensor, BN prologue or None) of a block's input."""
        if b.x_t is not None:
            return b.x_t.out, None
        return b.x_conv.z, b.x_bn

    def _build_forward(self):
        lib = self.lib
        ops = [self._conv_fwd(self.stem, self.feat, None, self.stem_bn)]
        for b in self.blocks:
            x, pro = self._block_input(b)
            ops.append(self._conv_fwd(b.c1, x, pro, b.b1))
            ops.append(self._conv_fwd(b.c2, b.c1.z, b.b1, b.b2))
            pixels = b.c2.N * b.c2.Hy * b.c2.Wy
            if b.cd is not None:
                ops.append(self._conv_fwd(b.cd, x, pro, b.bd))
                jargs = (b.out.data_ptr(), b.c2.z.data_ptr(), b.b2.scale.data_ptr(), b.b2.shift.data_ptr(), b.cd.z.data_ptr(), b.bd.scale.data_ptr(),
                         b.bd.shift.data_ptr(), 1, b.c2.Cy, pixels)
            elif pro is not None:
                jargs = (b.out.data_ptr(), b.c2.z.data_ptr(), b.b2.scale.data_ptr(), b.b2.shift.data_ptr(), x.data_ptr(), pro.scale.data_ptr(),
                         pro.shift.data_ptr(), 2, b.c2.Cy, pixels)
            else:
                jargs = (b.out.data_ptr(), b.c2.z.data_ptr(), b.b2.scale.data_ptr(), b.b2.shift.data_ptr(), x.data_ptr(), None, None, 0, b.c2.Cy, pixels)
            ops.append(lambda st, a=jargs: self._check(lib.bc_pn_join(*a, st), "pn_join"))
        ops.append(self._conv_fwd(self.head0, self.blocks[-1].out, None, self.head0_bn))
        ops.append(self._conv_fwd(self.head1, self.head0.z, self.head0_bn, self.head1_bn))
        hargs = (self.logits.data_ptr(), self.head1.z.data_ptr(), self.head1_bn.scale.data_ptr(), self.head1_bn.shift.data_ptr(),
                 self.P.data_ptr() + 4 * self.off_last_w, self.P.data_ptr() + 4 * self.off_last_b, self.N, self.last_Hi, self.last_Wi, self.last_C)
        ops.append(lambda st: self._check(lib.bc_pn_head_fwd(*hargs, st), "pn_head_fwd"))
        return ops

    def _bn_bwd(self, gz, g, conv: _Conv, bn: _BN, mask):
        """ops: gz = d loss / d conv.z through BN (+ ReLU: mask None -> own output; tensor -> that map > 0; False -> no ReLU)."""
        lib, G, P = self.lib, self.G, self.P
        mode, mptr = (1, None) if mask is None else ((0, None) if mask is False else (2, mask.data_ptr()))
        args = (gz.data_ptr(), G.data_ptr() + 4 * bn.off_g, G.data_ptr() + 4 * bn.off_b, self.bwd_part.data_ptr(), self.bwd_coef.data_ptr(), g.data_ptr(),
                conv.z.data_ptr(), mptr, mode, bn.scale.data_ptr(), bn.shift.data_ptr(), bn.mean.data_ptr(), bn.invstd.data_ptr(), P.data_ptr() + 4 * bn.off_g,
                bn.C, conv.N * conv.Hy * conv.Wy)
        return lambda st: self._check(lib.bc_pn_bn_bwd(*args, st), "pn_bn_bwd")

    def _wgrad(self, c: _Conv, x, pro: Optional[_BN], gz):
        lib = self.lib
        sc, sh = (pro.scale.data_ptr(), pro.shift.data_ptr()) if pro is not None else (None, None)
        # (dw = None: the partial copies stay in this conv's slice of the workspace; the step's last launch sums them, bc_pn_update)
        args = (None, self.wg_ws.data_ptr() + 4 * c.ws_off, c.groups * c.numel, x.data_ptr(), gz.data_ptr(), c.N, c.Hx, c.Wx, c.Cxp, c.Hy, c.Wy, c.Cy,
                c.ks, c.stride, sc, sh, 1 if pro is not None else 0)
        return lambda st: self._check(lib.bc_pn_wgrad_nhwc(*args, st), "pn_wgrad_nhwc")

    def _dgrad(self, c: _Conv, out, gz, add=None, add_mask=None, accumulate=0):
        lib = self.lib
        args = (out.data_ptr(), gz.data_ptr(), self.WT.data_ptr() + 4 * c.off_t, c.N, c.Hx, c.Wx, c.Cxp, c.Hy, c.Wy, c.Cy, c.ks, c.stride, 1, None, None, 0,
                add.data_ptr() if add is not None else None, add_mask.data_ptr() if add_mask is not None else None, accumulate, None, 0, BWD_PRECISION)
        return lambda st: self._check(lib.bc_pn_conv_nhwc(*args, st), "pn_conv_nhwc(dgrad)")

    def _build_step(self):
        lib = self.lib
        ops = []
        free = list(self.gbuf)
        take = lambda: free.pop()
        release = lambda b: free.append(b)

        # seed: reward + d loss / d logits (the information-gain map is produced by an eager launch right before the replay)
        self._seed_slot = len(ops)
        ops.append(None)      # filled by _bind_seed once the task output's geometry is known
        # last head stage
        gA = take()
        h1, b1 = self.head1, self.head1_bn
        a = (gA.data_ptr(), self.G.data_ptr() + 4 * self.off_last_w, self.G.data_ptr() + 4 * self.off_last_b, self.gl.data_ptr(), h1.z.data_ptr(),
             b1.scale.data_ptr(), b1.shift.data_ptr(), self.P.data_ptr() + 4 * self.off_last_w, self.N, self.last_Hi, self.last_Wi, self.last_C)
        ops.append(lambda st, a=a: self._check(lib.bc_pn_head_bwd(*a, st), "pn_head_bwd"))
        gZ = take()
        ops.append(self._bn_bwd(gZ, gA, h1, b1, None))
        release(gA)
        ops.append(self._wgrad(h1, self.head0.z, self.head0_bn, gZ))
        gA = take()
        ops.append(self._dgrad(h1, gA, gZ))
        release(gZ)
        gZ = take()
        ops.append(self._bn_bwd(gZ, gA, self.head0, self.head0_bn, None))
        release(gA)
        ops.append(self._wgrad(self.head0, self.blocks[-1].out, None, gZ))
        gOut = take()
        ops.append(self._dgrad(self.head0, gOut, gZ))
        release(gZ)
        for b in reversed(self.blocks):
            x, pro = self._block_input(b)
            gZ2 = take()
            ops.append(self._bn_bwd(gZ2, gOut, b.c2, b.b2, b.out))
            ops.append(self._wgrad(b.c2, b.c1.z, b.b1, gZ2))
            gMid = take()
            ops.append(self._dgrad(b.c2, gMid, gZ2))
            release(gZ2)
            gZ1 = take()
            ops.append(self._bn_bwd(gZ1, gMid, b.c1, b.b1, None))
            release(gMid)
            ops.append(self._wgrad(b.c1, x, pro, gZ1))
            gIn = take()
            if b.cd is None:
                ops.append(self._dgrad(b.c1, gIn, gZ1, add=gOut, add_mask=b.out))
                release(gZ1)
            else:
                ops.append(self._dgrad(b.c1, gIn, gZ1))
                release(gZ1)
                gZd = take()
                ops.append(self._bn_bwd(gZd, gOut, b.cd, b.bd, b.out))
                ops.append(self._wgrad(b.cd, x, pro, gZd))
                ops.append(self._dgrad(b.cd, gIn, gZd, accumulate=1))
                release(gZd)
            release(gOut)
            gOut = gIn
        gZ = take()
        ops.append(self._bn_bwd(gZ, gOut, self.stem, self.stem_bn, None))
        release(gOut)
        ops.append(self._wgrad(self.stem, self.feat, None, gZ))
        release(gZ)
        self._opt_slot = len(ops)
        ops.append(None)      # gradient sums + RMSprop with the optimizer's current hyper-parameters + export, one launch (_bind_optimizer)
        return ops

    def _bind_optimizer(self):
        g = self.optimizer.param_groups[0]
        hyper = (float(g["lr"]), float(g["alpha"]), float(g["eps"]), float(g["weight_decay"]), float(g["momentum"]))
        if hyper != self._hyper:
            def update(st, hyper=hyper):      # (the segment table is looked up at run time: it is rebuilt when the module's tensors move)
                self._check(self.lib.bc_pn_update(self.P.data_ptr(), self.G.data_ptr(), self.SQ.data_ptr(), self.MOM.data_ptr(), self.WT.data_ptr(),
                                                  self.wg_ws.data_ptr(), self.segs.data_ptr(), self.n_segs, *hyper, st), "pn_update")
            self._step_ops[self._opt_slot] = update
            self._hyper = hyper
            self._step_graph = None

    def _bind_seed(self, hq, wq, target, gamma):
        key = (hq, wq, float(target), float(gamma))
        if getattr(self, "_seed_key", None) != key:
            if self.ig is None or (self._hq, self._wq) != (hq, wq):
                self.ig = torch.zeros((self.N, 1, hq, wq), dtype=torch.float32, device=self.dev)
                self._hq, self._wq = hq, wq
            a = (self.gl.data_ptr(), self.loss.data_ptr(), self.reward.data_ptr(), self.logits.data_ptr(), self.grid_u8.data_ptr(), self.ig.data_ptr(),
                 self.cost.data_ptr(), 0.0, float(target), float(gamma), self.N, hq, wq, self.GH, self.GW)
            self._step_ops[self._seed_slot] = lambda st, a=a: self._check(self.lib.bc_pn_reward_seed(*a, st), "pn_reward_seed")
            self._seed_key = key
            self._step_graph = None

    # ------------------------------------------------------------------------------------------------------------ running
    def _run(self, ops, which):
        """One eager pass first (lazy code-object loads), then capture once and replay."""
        graph, warm = (self._fwd_graph, self._fwd_warm) if which == "fwd" else (self._step_graph, self._step_warm)
        if not USE_GRAPH or warm < 1 or torch.cuda.is_current_stream_capturing():
            st = self._stream()
            for f in ops:
                f(st)
            if which == "fwd":
                self._fwd_warm += 1
            else:
                self._step_warm += 1
            return
        if graph is None:
            torch.cuda.synchronize(self.dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                st = self._stream()
                for f in ops:
                    f(st)
            if which == "fwd":
                self._fwd_graph = graph
            else:
                self._step_graph = graph
        graph.replay()

    def features(self, policy_meta: Dict) -> bool:
        """Gather the policy input into ``self.feat`` (channels-last, 32 channels); False when a source is not usable."""
        frame, state, rep, grid = (policy_meta["inputs"], policy_meta["frame_state"], policy_meta.get("output_repr"), policy_meta.get("grid"))
        srcs_t = (frame, state, rep, grid)
        if any(t is None for t in srcs_t):
            return False
        key = tuple((t.shape, t.stride(), t.dtype) for t in srcs_t)
        desc = self._feat_desc.get(key)
        if desc is None:
            # geometry of the four sources (element strides, sizes, dtype codes, source-index scales): built once per layout, only the addresses
            # change from frame to frame
            srcs = self.net.feature_sources(policy_meta)
            if srcs is None:
                return False
            srcs, h, w = srcs
            assert (h, w) == (self.h, self.w)
            from blockcopy.backend import _DTYPE_CODE

            strides = (ctypes.c_longlong * 16)()
            dims = (ctypes.c_int * 16)()
            scales = (ctypes.c_float * 12)()
            for k, (t, sh, sw, off) in enumerate(srcs):
                code = 3 if t.dtype in (torch.bool, torch.uint8) else _DTYPE_CODE[t.dtype]
                strides[4 * k:4 * k + 4] = list(t.stride())
                dims[4 * k:4 * k + 4] = [t.shape[1], t.shape[2], t.shape[3], code]
                scales[3 * k:3 * k + 3] = [float(sh), float(sw), float(off)]
            desc = self._feat_desc[key] = ((ctypes.c_void_p * 4)(), strides, dims, scales)
        ptrs, strides, dims, scales = desc
        if not all(t.is_cuda for t in srcs_t):
            return False
        for k, t in enumerate(srcs_t):
            ptrs[k] = t.data_ptr()
        self._check(self.lib.bc_pn_features_nhwc(self.feat.data_ptr(), self.N, self.h, self.w, CPAD, ptrs, strides, dims, scales, self._stream()), "pn_features_nhwc")
        return True

    def decision_probs(self, grid: torch.Tensor):
        """(probs, log_probs) (N,1,GH,GW) of the decided grid under Bernoulli(logits of the LAST forward): one launch (bc_pn_probs)."""
        probs = torch.empty((self.N, 1, self.GH, self.GW), dtype=torch.float32, device=self.dev)
        logp = torch.empty_like(probs)
        g = grid.reshape(-1)
        g = g.view(torch.uint8) if g.dtype == torch.bool else g.to(torch.uint8)
        with torch.cuda.device(self.dev):
            self._check(self.lib.bc_pn_probs(probs.data_ptr(), logp.data_ptr(), self.logits.data_ptr(), g.data_ptr(), self.n_total, self._stream()), "pn_probs")
        return probs, logp

    @torch.no_grad()
    def forward(self, policy_meta: Dict) -> Optional[torch.Tensor]:
        """Tile logits (N, 1, GH, GW) -- a view of a static buffer, valid until the next forward -- or None (unsupported sources)."""
        with torch.cuda.device(self.dev):
            self._refresh_if_needed()
            if not self.features(policy_meta):
                return None
            self._run(self._fwd_ops, "fwd")
        self.forwards += 1
        return self.logits.view(self.N, 1, self.GH, self.GW)

    def forward_on(self, feat_nhwc: torch.Tensor) -> torch.Tensor:
        """Forward from a prepared input (N, h, w, 32) (tests)."""
        with torch.cuda.device(self.dev), torch.no_grad():
            self._refresh_if_needed()
            self.feat.copy_(feat_nhwc)
            self._run(self._fwd_ops, "fwd")
        self.forwards += 1
        return self.logits.view(self.N, 1, self.GH, self.GW)

    @torch.no_grad()
    def step(self, grid: torch.Tensor, outputs, outputs_prev, cost, target: float, gamma: float, scale_factor: float = 0.25, ig: torch.Tensor = None):
        """One REINFORCE update on the frame whose logits the LAST forward produced.  ``cost``: running execution rate (float, or a
        float64 0-d device tensor).  The information gain is computed here from the two task outputs (semantic segmentation: (N,C,H,W)
        logit maps of one layout) unless the caller hands over a precomputed map ``ig`` (N,1,hq,wq).  Returns (information_gain, loss (1,)) --
        views of static buffers."""
        from blockcopy.backend import _DTYPE_CODE

        if ig is None:
            assert outputs.shape == outputs_prev.shape and outputs.dtype == outputs_prev.dtype and outputs.dim() == 4
            assert outputs.stride() == outputs_prev.stride(), "current and previous task output must share a layout"
            N, C, H, W = outputs.shape
            hq, wq = int(np.floor(H * scale_factor)), int(np.floor(W * scale_factor))
        else:
            assert ig.dim() == 4 and ig.shape[0] == self.N and ig.shape[1] == 1
            hq, wq = ig.shape[2:]
        with torch.cuda.device(self.dev):
            self._refresh_if_needed()
            self._bind_seed(hq, wq, target, gamma)
            self._bind_optimizer()
            st = self._stream()
            if ig is not None:
                self.ig.copy_(ig)
            else:
                sn, sc, sh, sw = outputs.stride()
                self._check(self.lib.bc_pn_infogain(self.ig.data_ptr(), outputs.data_ptr(), outputs_prev.data_ptr(), _DTYPE_CODE[outputs.dtype], N, C, H, W, sn, sc,
                                                    sh, sw, hq, wq, 1.0 / scale_factor, 1.0 / scale_factor, st), "pn_infogain")
            g = grid.reshape(-1)
            if g.data_ptr() != self.grid_u8.data_ptr():
                self.grid_u8.copy_(g.view(torch.uint8) if g.dtype == torch.bool else g.to(torch.uint8))
            if isinstance(cost, torch.Tensor):
                self.cost.copy_(cost.reshape(1))
            else:
                self.cost.fill_(float(cost))
            self._run(self._step_ops, "step")
        self.steps += 1
        self._seen_versions = self._versions()      # (the export kernel writes the parameters without touching their version counters)
        return self.ig, self.loss


def try_build(net, optimizer, frame_shape, device):
    """NativePolicyNet for this module / frame geometry, or None (with the reason recorded on the net) when not applicable."""
    if not ENABLED:
        return None
    why = supported(net, optimizer)
    if why is None:
        try:
            return NativePolicyNet(net, optimizer, frame_shape, device)
        except (RuntimeError, AttributeError) as e:     # e.g. a checker backend without the bc_pn_* entry points
            why = str(e)
    net._native_unsupported = why
    return None
