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


# ----------------------------------------------------------------------------------------------------------------
# Pedestron / mmdet side of the path (BASELINE config C5): the reference's OWN detector modules, loaded by path.
PED_ROOT = os.path.join(REF_ROOT, "Pedestron")


def _stub_pkg(name, path=None, **attrs):
    """Register an empty package object under ``name`` so that importing a real submodule below it does NOT execute the
    package's own ``__init__.py`` (mmdet's import everything: DCN / RoI extensions, datasets, pycocotools ...).  ``path`` = the
    real directory: submodules requested later are found there and are the reference's files, unmodified."""
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = [path] if path else []
        m.__package__ = name
        sys.modules[name] = m
        if "." in name:
            parent, leaf = name.rsplit(".", 1)
            setattr(sys.modules[parent], leaf, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


class AttrDict(dict):
    """What mmcv.Config hands the head as ``test_cfg`` (attribute access + dict methods; csp_head.py:257,276-280)."""

    __getattr__ = dict.__getitem__


def load_reference_csp(ref):
    """Import the reference's CSP detector stack -- detectors/{base,single_stage,csp,csp_blockcopy}.py, backbones/resnet.py,
    necks/csp_neck.py, anchor_heads/csp_head.py, models/utils/*, models/{registry,builder}.py, mmdet/utils/registry.py,
    core/bbox/transforms.py, core/post_processing/bbox_nms.py, core/utils/misc.py, core/fp16/{decorators,utils}.py,
    ops/nms/nms_wrapper.py -- as the reference's files under their real module names.  What is NOT the reference:

    * ``mmcv`` (not installed): ``mmcv.is_str`` and the four ``mmcv.cnn`` initialisers (weights are overwritten by name-seeded
      values anyway), ``mmcv.runner.load_checkpoint`` (never called: pretrained=None);
    * ``cv2`` / ``pycocotools.mask``: imported by those files, never called on this path;
    * ``mmdet.ops`` compiled extensions: DeformConv & co. are placeholders (dcn=None in the C5 config); the two NMS extension
      modules ``nms_cuda`` / ``nms_cpu`` that nms_wrapper.py dispatches to are the oracle's restatement of nms_kernel.cu
      (oracle.c_nms) -- the CUDA rule (suppress at IoU > thr), because the reference runs this path on the GPU;
    * package ``__init__`` files of mmdet / mmdet.models / mmdet.core (they import the whole framework).

    ``ref`` = load_reference() (the detector imports the reference's ``blockcopy``).  Returns a namespace."""
    import importlib

    assert ref.bc.__file__.startswith(REF_ROOT)
    mm = os.path.join(PED_ROOT, "mmdet")
    for name in ("cv2", "pycocotools", "pycocotools.mask"):
        if name not in sys.modules:
            _stub_pkg(name)

    # --- mmcv stand-ins
    import torch.nn as nn

    def constant_init(module, val, bias=0):
        nn.init.constant_(module.weight, val)
        if hasattr(module, "bias") and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    _stub_pkg("mmcv", is_str=lambda x: isinstance(x, str))
    _stub_pkg("mmcv.runner", load_checkpoint=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no checkpoints here")))

    # --- mmdet package skeleton (real directories, stub __init__)
    _stub_pkg("mmdet", mm)
    utils = importlib.import_module("mmdet.utils")                       # real: utils/{__init__,registry}.py
    _stub_pkg("mmdet.models", os.path.join(mm, "models"))
    # mmcv.cnn's initialisers (third-party, absent): same signatures; they only matter until the name-seeded values are loaded.
    # (The reference's own copies in models/utils/weight_init.py cannot stand in: they write module.bias without a None check.)
    def _init(fill):
        def init(module, *a, bias=0, **k):
            fill(module.weight, *a, **{kk: v for kk, v in k.items() if kk != "distribution"})
            if getattr(module, "bias", None) is not None:
                nn.init.constant_(module.bias, bias)
        return init

    _stub_pkg("mmcv.cnn", normal_init=_init(lambda w, mean=0, std=1: nn.init.normal_(w, mean, std)),
              xavier_init=_init(lambda w, gain=1: nn.init.xavier_uniform_(w, gain=gain)),
              kaiming_init=_init(lambda w, mode="fan_out", nonlinearity="relu": nn.init.kaiming_normal_(w, mode=mode, nonlinearity=nonlinearity)),
              constant_init=constant_init)
    mutils = importlib.import_module("mmdet.models.utils")               # real: conv_module / norm / scale / conv_ws / weight_init
    registry = importlib.import_module("mmdet.models.registry")          # real
    builder = importlib.import_module("mmdet.models.builder")            # real

    placeholder = type("NotOnThisPath", (), {"__init__": lambda self, *a, **k: (_ for _ in ()).throw(RuntimeError("mmdet.ops extension"))})
    _stub_pkg("mmdet.ops", os.path.join(mm, "ops"), DeformConv=placeholder, ModulatedDeformConv=placeholder, ContextBlock=placeholder)
    _stub_pkg("mmdet.models.plugins", GeneralizedAttention=placeholder)
    _stub_pkg("mmdet.ops.nms", os.path.join(mm, "ops", "nms"))

    def oracle_nms(dets, thr):
        return torch.from_numpy(O.c_nms(dets.detach().float().cpu().numpy(), float(thr))).to(dets.device)

    for ext in ("nms_cuda", "nms_cpu"):
        _stub_pkg(f"mmdet.ops.nms.{ext}", nms=oracle_nms)
    _stub_pkg("mmdet.ops.nms.soft_nms_cpu", soft_nms_cpu=None)
    nms_wrapper = importlib.import_module("mmdet.ops.nms.nms_wrapper")   # real dispatcher

    # --- mmdet.core: the real files the path calls, gathered under the names the modules import them by
    _stub_pkg("mmdet.core", os.path.join(mm, "core"))
    for sub in ("bbox", "post_processing", "utils", "fp16"):
        _stub_pkg(f"mmdet.core.{sub}", os.path.join(mm, "core", sub))
    transforms = importlib.import_module("mmdet.core.bbox.transforms")
    bbox_nms = importlib.import_module("mmdet.core.post_processing.bbox_nms")
    misc = importlib.import_module("mmdet.core.utils.misc")
    decorators = importlib.import_module("mmdet.core.fp16.decorators")
    core = sys.modules["mmdet.core"]
    for src, names in ((transforms, ("bbox2result", "csp_height2bbox", "csp_heightwidth2bbox")), (bbox_nms, ("multiclass_nms",)),
                       (misc, ("multi_apply", "tensor2imgs")), (decorators, ("auto_fp16", "force_fp32"))):
        for n in names:
            setattr(core, n, getattr(src, n))
    core.get_classes = lambda *a, **k: ()

    # --- the model files
    for pkg in ("backbones", "necks", "anchor_heads", "detectors"):
        _stub_pkg(f"mmdet.models.{pkg}", os.path.join(mm, "models", pkg))
    resnet = importlib.import_module("mmdet.models.backbones.resnet")
    neck = importlib.import_module("mmdet.models.necks.csp_neck")
    head = importlib.import_module("mmdet.models.anchor_heads.csp_head")
    det = importlib.import_module("mmdet.models.detectors.csp_blockcopy")
    for m in (utils, mutils, registry, builder, nms_wrapper, transforms, bbox_nms, misc, decorators, resnet, neck, head, det):
        assert m.__file__.startswith(PED_ROOT), m.__file__
    return types.SimpleNamespace(resnet=resnet, neck=neck, head=head, det=det, csp=sys.modules["mmdet.models.detectors.csp"],
                                 builder=builder, transforms=transforms, bbox_nms=bbox_nms, nms_wrapper=nms_wrapper, AttrDict=AttrDict)


def csp_r50_config(block_size, policy="all"):
    """Constructor arguments of the reference's C5 model = Pedestron/configs/elephant/cityperson/csp_r50_clip_blockcopy_030.py:3-70
    (model / test_cfg dicts; pretrained=None: no checkpoints here; policy replaced by forced grids in the fixtures)."""
    model = dict(
        blockcopy_settings=dict(block_policy=policy, block_num_classes=1, block_optim_lr=1e-4, block_optim_wd=1e-4, block_optim_momentum=0,
                                block_target=0.3, block_complexity_weight=5, block_size=block_size, block_train_interval=4,
                                block_cost_momentum=0.9, block_policy_verbose=False),
        pretrained=None,
        backbone=dict(type="ResNet", depth=50, num_stages=4, strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), out_indices=(1, 2, 3),
                      frozen_stages=-1, norm_eval=False, style="pytorch"),
        neck=dict(type="CSPNeck", in_channels=[512, 1024, 2048], out_channels=256, start_level=0, add_extra_convs=True,
                  extra_convs_on_inputs=False, num_outs=5, relu_before_extra_convs=True),
        bbox_head=dict(type="CSPHead", num_classes=2, in_channels=768, stacked_convs=1, feat_channels=256, strides=[4],
                       loss_cls=AttrDict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=0.01),
                       loss_bbox=AttrDict(type="IoULoss", loss_weight=1), loss_offset=AttrDict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=0.1)))
    test_cfg = AttrDict(nms_pre=1000, min_bbox_size=0, score_thr=0.1, nms=AttrDict(type="nms", iou_thr=0.5), max_per_img=100)
    return model, test_cfg
