"""Load the REFERENCE Python package in this container so golden vectors can be generated.

Test infrastructure, only usable where /root/reference exists (never on the GPU box; nothing
from the reference is copied into the repo).  Recipe = SURVEY.md Appendix A:

1. register a stub ``cupy`` module (only ``cupy.memoize`` / ``cupy.cuda`` are touched at import,
   blockcopy/blockcopy/utils/cuda.py:6,25,30);
2. replace the ``forward`` of the four CuPy-backed autograd Functions with the CPU oracle
   (oracle/oracle.py ``c_*``), keeping the reference's own host wrappers' argument contract;
3. replace ``to_tensorwrapper`` (asserts ``is_cuda``, core/tensorwrapper.py:35).

Everything else -- get_grid_mappings (TorchScript), BlockFeatures FIFO, TensorWrapper
``__torch_function__`` routing, BlockCopyModel state machine, policies, SwiftNet, BN folding --
is the reference's own code executing.
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import torch

REF_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import oracle as O  # noqa: E402

CALL_LOG = None  # set to a list to record every kernel stand-in call (op-level fixtures)


def _log(name, **kw):
    if CALL_LOG is not None:
        CALL_LOG.append((name, {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in kw.items()}))


def load_reference():
    """Import the reference's ``blockcopy`` + SwiftNet modules with oracle kernel stand-ins.  Returns a namespace."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (fixtures can only be regenerated in the build container)")
    if "cupy" not in sys.modules:
        cupy = types.ModuleType("cupy")
        cupy.memoize = lambda for_each_device=False: (lambda f: f)
        cupy.cuda = types.SimpleNamespace()
        sys.modules["cupy"] = cupy
    for p in (os.path.join(REF_ROOT, "blockcopy"), os.path.join(REF_ROOT, "semantic_segmentation")):
        if p not in sys.path:
            sys.path.insert(0, p)
    # our own package must not shadow the reference's import name here
    for name in list(sys.modules):
        if name == "blockcopy" or name.startswith("blockcopy."):
            f = getattr(sys.modules[name], "__file__", "") or ""
            if not f.startswith(REF_ROOT):
                raise RuntimeError("a non-reference `blockcopy` is already imported in this process")
    warnings.filterwarnings("ignore", message=".*__torch_function__.*")
    import blockcopy as ref_bc
    import blockcopy.core.tensorwrapper as ref_tw
    import blockcopy.utils.block_funcs as ref_bf
    import blockcopy.utils.blockpad as ref_bp

    assert ref_bc.__file__.startswith(REF_ROOT), ref_bc.__file__

    def split_fwd(ctx, blocks, image, mapping_exec, grid_idx):
        if len(mapping_exec) > 0:
            O.c_split(blocks, image, mapping_exec)
        _log("split", image=image, mapping_exec=mapping_exec, grid_idx=grid_idx, out=blocks)
        return blocks

    def combine_fwd(ctx, blocks, out, grid_idx, mapping_exec):
        before = out.detach().clone() if CALL_LOG is not None else None
        if len(mapping_exec) > 0:
            O.c_combine(blocks, out, mapping_exec)
        _log("combine", blocks=blocks, out_before=before, grid_idx=grid_idx, mapping_exec=mapping_exec, out=out)
        return out

    def transfer_fwd(ctx, data_transfer, prev_computed, prev_transfer, grid_idx_prev, transfer_map_prev, padding):
        # deterministic fill of the don't-care interior (the reference leaves torch.empty garbage there)
        data_transfer.fill_(float("nan"))
        if len(transfer_map_prev) > 0:
            O.c_transfer(data_transfer, prev_computed, prev_transfer, tuple(grid_idx_prev.shape), transfer_map_prev, padding)
        _log("transfer", prev_computed=prev_computed, prev_transfer=prev_transfer, grid_idx_prev=grid_idx_prev,
             transfer_idx=transfer_map_prev, padding=padding, out=data_transfer)
        return data_transfer

    def pad_fwd(ctx, data_exec, data_transfer, grid_idx, mapping_exec, pad):
        B, C, bs, _ = data_exec.shape
        out = torch.empty((B, C, bs + 2 * pad, bs + 2 * pad), dtype=data_exec.dtype)
        if len(mapping_exec) > 0:
            O.c_repad(out, data_exec, data_transfer, grid_idx, mapping_exec, pad)
        _log("pad", data_exec=data_exec, data_transfer=data_transfer, grid_idx=grid_idx, mapping_exec=mapping_exec, pad=pad, out=out)
        return out

    ref_bf.SplitFunction.forward = staticmethod(split_fwd)
    ref_bf.CombineFunction.forward = staticmethod(combine_fwd)
    ref_bf.TransferFunction.forward = staticmethod(transfer_fwd)
    ref_bp.BlockPadFunction.forward = staticmethod(pad_fwd)

    to_tw = lambda x: x.as_subclass(ref_tw.TensorWrapper)  # noqa: E731
    ref_bc.to_tensorwrapper = to_tw
    ref_tw.to_tensorwrapper = to_tw

    from lib.models.swiftnet import swiftnet as ref_swiftnet  # noqa
    from lib.models.swiftnet.backbones import resnet as ref_resnet  # noqa
    from lib.utils import bn_fusion as ref_bn_fusion  # noqa
    from blockcopy.policy import policy as ref_policy  # noqa

    return types.SimpleNamespace(bc=ref_bc, tw=ref_tw, bf=ref_bf, bp=ref_bp, swiftnet=ref_swiftnet,
                                 resnet=ref_resnet, bn_fusion=ref_bn_fusion, policy=ref_policy)


def make_forced_policy(ref, block_size, grids):
    """A reference ``Policy`` subclass that replays a list of grids (Appendix A step 6)."""

    class ForcedPolicy(ref.policy.Policy):
        def __init__(self):
            super().__init__(block_size=block_size, verbose=False)
            self._grids = list(grids)
            self._t = 0

        def forward(self, policy_meta):
            policy_meta["grid"] = self._grids[self._t].clone()
            self._t += 1
            return self.stats.add_policy_meta(policy_meta)

    return ForcedPolicy()
