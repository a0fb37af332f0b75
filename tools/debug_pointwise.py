#!/usr/bin/env python3
"""Debug aid: run a golden SwiftNet clip with every bc_conv1x1_nhwc / bc_conv3x3_ring launch re-checked against stock PyTorch
on the same operands (prologue -> conv -> epilogue), printing the launches that disagree.
usage: python tools/debug_pointwise.py [fixture.npz] [--conv native|auto]"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fixture", nargs="?", default="swiftnet_rn50_a.npz")
    ap.add_argument("--conv", default="native")
    ap.add_argument("--hooks", default="", help="comma list of module class names whose outputs are compared between deferred / immediate conv launches")
    a = ap.parse_args()
    import blockcopy
    import blockcopy.backend as bk
    from blockcopy.core import fusion
    from blockcopy.core.argparser import default_settings
    from bc_workloads import seeded
    from bc_workloads.bn_fold import fold_batchnorm
    from bc_workloads.swiftnet import build_swiftnet
    from common import make_forced_policy

    fusion.CONV_MODE = a.conv
    be = bk.get_backend()
    weights = {}
    orig_pack = be.pack_conv3x3_weights

    def pack(w):
        out = orig_pack(w)
        weights[out.data_ptr()] = w.detach().as_subclass(torch.Tensor)
        return out

    be.pack_conv3x3_weights = pack
    orig = be.conv1x1

    def affine(x, sc, sh, add, relu):
        y = x.float()
        if sc is not None:
            y = y * sc.view(1, -1, 1, 1)
        if sh is not None:
            y = y + sh.view(1, -1, 1, 1)
        if add is not None:
            y = y + add.float()
        return torch.relu(y) if relu else y

    def checked(data, wpk, cout, prologue=None, epilogue=None, cfg=None, stride=1):
        out = orig(data, wpk, cout, prologue, epilogue, cfg, stride)
        w = weights[wpk.data_ptr()]
        x = affine(data, prologue[0], prologue[1], None, prologue[2]) if prologue is not None else data.float()
        y = F.conv2d(x, w.float(), stride=stride)
        if epilogue is not None:
            y = affine(y, *epilogue)
        d = float((out.float() - y).abs().max()) if out.numel() else 0.0
        flag = "  <<<<<<" if not d <= 1e-3 else ""
        print(f"conv1x1 data {tuple(data.shape)} strides {data.stride()} cout {cout} stride {stride} cfg {cfg} pro {prologue is not None} "
              f"epi {None if epilogue is None else [e is not None and (e is True or e is not False) for e in epilogue]} maxdiff {d:.3e}{flag}", flush=True)
        return out

    be.conv1x1 = checked

    if a.hooks:
        be.conv1x1 = orig
        return compare_modes(a, blockcopy, fusion)
    gd = os.path.join(ROOT, "tests", "golden")
    G = np.load(os.path.join(gd, a.fixture))
    cfg = json.loads(bytes(G["cfg"]).decode())
    net = build_swiftnet(cfg["backbone"])
    net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()), strict=True)
    net.eval()
    model = blockcopy.BlockCopyModel(net, default_settings(block_policy="all", block_size=cfg["block_size"], block_graph=0))
    grids = [torch.from_numpy(G[f"grid{t}"]) for t in range(cfg["n_frames"])]
    model.policy = make_forced_policy(cfg["block_size"], grids)
    model = fold_batchnorm(model.cuda()).to(memory_format=torch.channels_last)
    with torch.no_grad():
        model.reset_temporal()
        for t in range(cfg["n_frames"]):
            print(f"---- frame {t} executed {int(grids[t].sum())}", flush=True)
            y = model(seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"])).cuda())
            print(f"frame {t}: logits maxdiff {float((y.cpu() - torch.from_numpy(G[f'logits{t}'])).abs().max()):.3e}", flush=True)


def compare_modes(a, blockcopy, fusion):
    from blockcopy.core.argparser import default_settings
    from blockcopy.core.tensorwrapper import TensorWrapper
    from bc_workloads import seeded
    from bc_workloads.bn_fold import fold_batchnorm
    from bc_workloads.swiftnet import build_swiftnet
    from common import make_forced_policy

    G = np.load(os.path.join(ROOT, "tests", "golden", a.fixture))
    cfg = json.loads(bytes(G["cfg"]).decode())
    names = set(a.hooks.split(","))

    def run(defer):
        fusion.DEFER_CONV = defer
        net = build_swiftnet(cfg["backbone"])
        net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()), strict=True)
        net.eval()
        model = blockcopy.BlockCopyModel(net, default_settings(block_policy="all", block_size=cfg["block_size"], block_graph=0))
        grids = [torch.from_numpy(G[f"grid{t}"]) for t in range(cfg["n_frames"])]
        model.policy = make_forced_policy(cfg["block_size"], grids)
        model = fold_batchnorm(model.cuda()).to(memory_format=torch.channels_last)
        rec, frame = [], [0]

        def plain(o):
            if isinstance(o, TensorWrapper):
                return o._plain().detach().clone().float().cpu()
            if isinstance(o, torch.Tensor):
                return o.detach().clone().float().cpu()
            if isinstance(o, (tuple, list)):
                return [plain(v) for v in o]
            return None

        for n, m in model.named_modules():
            if type(m).__name__ in names:
                m.register_forward_hook((lambda n_: lambda mod, inp, out: rec.append((frame[0], n_, plain(out))))(n))
        with torch.no_grad():
            model.reset_temporal()
            for t in range(cfg["n_frames"]):
                frame[0] = t
                y = model(seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"])).cuda())
                print(f"defer {defer} frame {t}: logits maxdiff {float((y.cpu() - torch.from_numpy(G[f'logits{t}'])).abs().max()):.3e}", flush=True)
        return rec

    A, B = run(True), run(False)

    def diff(x, y):
        if isinstance(x, list):
            return max([diff(u, v) for u, v in zip(x, y)] + [0.0])
        if x is None:
            return 0.0
        return float((x - y).abs().max()) if x.numel() else 0.0

    for (fa, na, oa), (fb, nb, ob) in zip(A, B):
        assert (fa, na) == (fb, nb), (fa, na, fb, nb)
        d = diff(oa, ob)
        if fa == 2:
            print(f"frame {fa} {na:40s} maxdiff {d:.3e}{'  <<<<' if d > 1e-3 else ''}")


if __name__ == "__main__":
    main()
