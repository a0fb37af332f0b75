"""Training-mode BatchNorm2d of the policy net through bc_bn_train_fwd.

The online-RL policies keep their small CNN in train() mode on EVERY frame (reference policy/policy.py: batch statistics of
the single frame, slow running statistics, online updates), so each frame pays ten batch-statistics BatchNorms.  The stock
op is three library kernels plus a momentum update and a counter increment per layer; ``PolicyBatchNorm2d`` computes the
same forward (optionally with the following ReLU) in two launches and hands the saved statistics to ATen's own
``native_batch_norm_backward`` for the training steps.  Parameter and buffer names are those of ``nn.BatchNorm2d``
(checkpoints load unchanged); anything the kernel does not cover (CPU tensors, eval mode, other dtypes or layouts) takes the
stock module path."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

FUSED_BN = os.environ.get("BLOCKCOPY_FUSED_BN", "1") != "0"
FUSED_BN_GRAD = os.environ.get("BLOCKCOPY_FUSED_BN_GRAD", "0") != "0"     # also under autograd (tests; slower end to end, see forward)


class _BNTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, batches, momentum, eps, relu):
        from blockcopy.backend import get_backend

        y, save_mean, save_invstd = get_backend().bn_train(x, weight, bias, running_mean, running_var, batches, momentum, eps, relu)
        ctx.save_for_backward(x, weight, save_mean, save_invstd, y if relu else None)
        ctx.eps, ctx.relu = eps, relu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, save_mean, save_invstd, y = ctx.saved_tensors
        if ctx.relu:
            gy = gy * (y > 0).to(gy.dtype)
        cl = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
        gy = gy.contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)      # the layout of x: no transposing backward
        gx, gw, gb = torch.ops.aten.native_batch_norm_backward(gy, x, weight, None, None, save_mean, save_invstd, True, ctx.eps,
                                                               [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]])
        return gx, gw, gb, None, None, None, None, None, None


class PolicyBatchNorm2d(nn.BatchNorm2d):
    def forward(self, x, relu: bool = False):
        # Only where no gradient is recorded (the per-frame decision forward, captured in a hipGraph: no host cost).  The training
        # forward keeps the stock op: measured on C3, the autograd.Function route is 0.1 ms/frame faster on the GPU but its Python
        # cost (ten custom Functions + allocations) makes the host-bound training frames slower (462-480 -> 385-417 fps).
        if (FUSED_BN and (FUSED_BN_GRAD or not torch.is_grad_enabled()) and self.training and self.track_running_stats and self.momentum is not None and self.affine and x.is_cuda
                and x.dtype == torch.float32 and x.dim() == 4 and self.weight.dtype == torch.float32):
            from blockcopy.backend import get_backend

            be = get_backend()
            if hasattr(be, "bn_train") and be.bn_train_supported(x):
                return _BNTrain.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked,
                                      self.momentum, self.eps, relu)
        y = super().forward(x)
        return F.relu(y) if relu else y
