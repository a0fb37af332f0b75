"""Multiply-accumulate counter with the reference's semantics (Pedestron/tools/flopscounter.py,
the ptflops derivative both reference drivers use: test_swiftnet.py:239-244, tools/test_city_person.py).

What it counts -- and only that (flopscounter.py:423-459): Conv1d/2d/3d, ConvTranspose2d and Linear modules, through
forward hooks, from the shapes of the tensors that actually flow.  A conv costs
``prod(kernel) * Cin * Cout / groups`` per OUTPUT position (+ Cout per position with a bias) and the number of positions is
``output.shape[0] * H_out * W_out`` (:341-373): under block-copy execution ``output.shape[0]`` is the number of EXECUTED
tiles, so the count shrinks with the execution rate -- that is how the reference's GMACs figures are produced.
``compute_average_flops_cost()`` divides by the number of frames that went through the top-level module."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


def _conv_macs(m, inputs, output):
    per_position = int(np.prod(m.kernel_size)) * m.in_channels * (m.out_channels // m.groups)
    positions = int(output.shape[0]) * int(np.prod(output.shape[2:]))
    return per_position * positions + (m.out_channels * positions if m.bias is not None else 0)


def _deconv_macs(m, inputs, output):
    x = inputs[0]
    kh, kw = m.kernel_size
    macs = kh * kw * m.in_channels * (m.out_channels // m.groups) * int(x.shape[0]) * int(x.shape[2]) * int(x.shape[3])
    if m.bias is not None:
        macs += m.out_channels * int(x.shape[0]) * int(output.shape[2]) * int(output.shape[2])   # (height twice: reference :331)
    return macs


def _linear_macs(m, inputs, output):
    return int(np.prod(inputs[0].shape)) * int(output.shape[-1])


_COUNTERS = {nn.Conv1d: _conv_macs, nn.Conv2d: _conv_macs, nn.Conv3d: _conv_macs, nn.ConvTranspose2d: _deconv_macs, nn.Linear: _linear_macs}


class GMACsCounter:
    """``with GMACsCounter(model) as c: ...run frames...; c.compute_average_flops_cost()``."""

    def __init__(self, model: nn.Module):
        self.model = model
        self.macs = 0
        self.frames = 0
        self.per_module = {}
        self._handles = []

    def start_flops_count(self):
        if self._handles:
            return self
        self._handles.append(self.model.register_forward_hook(self._count_frames))
        for name, m in self.model.named_modules():
            fn = _COUNTERS.get(type(m))
            if fn is not None:
                self._handles.append(m.register_forward_hook(self._make_hook(name, fn)))
        return self

    def stop_flops_count(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def reset_flops_count(self):
        self.macs, self.frames, self.per_module = 0, 0, {}

    def _count_frames(self, module, inputs, output):
        self.frames += len(inputs[0]) if len(inputs) > 0 else 1

    def _make_hook(self, name, fn):
        def hook(module, inputs, output):
            n = int(fn(module, inputs, output))
            self.macs += n
            self.per_module[name] = self.per_module.get(name, 0) + n
        return hook

    def compute_average_flops_cost(self):
        """(MACs per frame, number of frames), or 0 before any frame -- the reference's return convention
        (flopscounter.py:154-173; the drivers print ``[0] / 1e9`` as GMACs, test_swiftnet.py:174)."""
        if self.frames == 0:
            return 0
        return self.macs / self.frames, self.frames

    def compute_total_flops_cost(self):
        """(MACs per frame, per-module MACs per frame, number of frames): despite its name the reference's method also
        divides by the number of frames (flopscounter.py:175-208)."""
        n = max(1, self.frames)
        return self.macs / n, {k: v / n for k, v in self.per_module.items()}, self.frames

    __enter__ = start_flops_count

    def __exit__(self, *exc):
        self.stop_flops_count()
