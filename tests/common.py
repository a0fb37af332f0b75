"""Helpers shared by CPU and GPU tests."""
import json
import os

import numpy as np
import torch


def load_golden(golden_dir, name):
    G = np.load(os.path.join(golden_dir, name))
    cfg = json.loads(bytes(G["cfg"]).decode()) if "cfg" in G.files else None
    return G, cfg


def golden_logit_error(G, cfg, t, y):
    """max |y - golden logits of frame t|; full-size fixtures hold 8-strided lattices of the map instead of the map
    (oracle/gen_golden.py gen_swiftnet, ``subsample``) plus its absolute maximum."""
    y = y.detach().float().cpu()
    sub = cfg.get("subsample")
    if sub is None:
        return float((y - torch.from_numpy(G[f"logits{t}"])).abs().max())
    err = 0.0
    for o in sub["offsets"]:
        err = max(err, float((y[:, :, o::sub["step"], o::sub["step"]] - torch.from_numpy(G[f"logits{t}_o{o}"])).abs().max()))
    err = max(err, abs(float(y.abs().max()) - float(G[f"logits{t}_absmax"])))
    return err


def make_forced_policy(block_size, grids):
    """A Policy that replays a list of host grids (used to pin the execution masks of the golden clips)."""
    from blockcopy.policy.policy import Policy

    class ForcedPolicy(Policy):
        def __init__(self):
            super().__init__(block_size)
            self._grids = list(grids)
            self._t = 0

        def forward(self, policy_meta):
            self.publish(policy_meta, self._grids[self._t])
            self._t += 1
            return self.stats.add_policy_meta(policy_meta)

    return ForcedPolicy()


def build_wrapped_swiftnet(cfg, grids, device, engine, graph=0):
    import blockcopy
    from blockcopy.core import tensorwrapper as tw
    from blockcopy.core.argparser import default_settings
    from bc_workloads import seeded
    from bc_workloads.bn_fold import fold_batchnorm
    from bc_workloads.swiftnet import build_swiftnet

    tw.set_engine(engine)
    net = build_swiftnet(cfg["backbone"])
    net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()), strict=True)
    net.eval()
    model = blockcopy.BlockCopyModel(net, default_settings(block_policy="all", block_size=cfg["block_size"], block_graph=graph))
    model.policy = make_forced_policy(cfg["block_size"], grids)
    model = fold_batchnorm(model.to(device))
    model.reset_temporal()
    return model


def run_golden_clip(G, cfg, device, engine, graph=0, repeats=1):
    """Replays a golden SwiftNet clip; returns per-frame max abs error of the logits (and of frame_state if stored)."""
    from bc_workloads import seeded

    grids = [torch.from_numpy(G[f"grid{t}"]) for t in range(cfg["n_frames"])]
    model = build_wrapped_swiftnet(cfg, grids * repeats, device, engine, graph)
    errs, fs_errs = [], []
    with torch.no_grad():
        for rep in range(repeats):
            if rep:
                model.reset_temporal()   # a new clip over the same persistent buffers / captured graphs
            for t in range(cfg["n_frames"]):
                x = seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"])).to(device)
                y = model(x)
                errs.append(golden_logit_error(G, cfg, t, y))
                if f"frame_state{t}" in G.files:
                    fs_errs.append(float((model.policy_meta["frame_state"].cpu() - torch.from_numpy(G[f"frame_state{t}"])).abs().max()))
    return errs, fs_errs


def run_golden_detector_clip(G, cfg, modules, device, engine, graph=0, channels_last=False, fold_bn=False, repeats=1):
    """Replays a golden detector clip (tests/golden/tinycsp.npz, csp_r50.npz: head maps the REFERENCE TensorWrapper
    produced for the same modules) through this repo's CSPBlockCopy manager; returns per-frame max abs errors of the
    three head maps relative to max(1, |golden|max), and of frame_state (absolute)."""
    from blockcopy.core import tensorwrapper as tw
    from blockcopy.core.argparser import default_settings
    from bc_workloads import seeded
    from bc_workloads.bn_fold import fold_batchnorm
    from bc_workloads.csp import CSPBlockCopy

    tw.set_engine(engine)
    grids = [torch.from_numpy(G[f"grid{t}"]) for t in range(cfg["n_frames"])]
    det = CSPBlockCopy(default_settings(block_policy="all", block_size=cfg["block_size"], block_graph=graph), results="device", arch=modules)
    det.policy = make_forced_policy(cfg["block_size"], grids * repeats)
    det.eval()
    det = det.to(device)
    if fold_bn:
        det = fold_batchnorm(det)
    if channels_last:
        det = det.to(memory_format=torch.channels_last)
    errs, fs_errs = [], []
    for rep in range(repeats):
        det.reset_temporal()
        for t in range(cfg["n_frames"]):
            x = seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"])).to(device)
            maps = det.forward_maps(x)
            e = 0.0
            for k, m in zip(("cls", "reg", "offset"), maps):
                want = torch.from_numpy(G[f"{k}{t}"])
                assert tuple(m.shape) == tuple(want.shape), (k, t, m.shape, want.shape)
                e = max(e, float((m.float().cpu() - want).abs().max()) / max(1.0, float(want.abs().max())))
            errs.append(e)
            if f"frame_state{t}" in G.files:
                fs_errs.append(float((det.policy_meta["frame_state"].cpu() - torch.from_numpy(G[f"frame_state{t}"])).abs().max()))
    return errs, fs_errs


def scenario_grids(N, GH, GW, seed):
    """The mask sequence of the golden clips (oracle/gen_golden.py scenario_grids): all, half, one, all-but-one, none, quarter."""
    from bc_workloads import seeded

    total = N * GH * GW
    return [torch.ones(N, 1, GH, GW, dtype=torch.bool),
            seeded.fixed_fraction_grid(seed + 1, N, GH, GW, total // 2),
            seeded.fixed_fraction_grid(seed + 2, N, GH, GW, 1),
            seeded.fixed_fraction_grid(seed + 3, N, GH, GW, total - 1),
            torch.zeros(N, 1, GH, GW, dtype=torch.bool),
            seeded.fixed_fraction_grid(seed + 5, N, GH, GW, total // 4)]


def nms_known_answers():
    """Hand-derived known answers for greedy NMS as Pedestron/mmdet/ops/nms/src/nms_kernel.cu defines it: IoU with the +1 pixel
    convention (devIoU, :12-21: width = right - left + 1), a box is suppressed by an earlier KEPT box when IoU > threshold
    (strict, :61), candidates in descending score order (:73-75).  Integer boxes, so every IoU below is an exact ratio.
    Each case: (dets (n,5) float32 [x1,y1,x2,y2,score], threshold, kept original indices ascending)."""
    f = lambda rows: np.array(rows, dtype=np.float32)
    A, B = [0, 0, 9, 9], [0, 0, 9, 4]          # areas 100 and 50 (+1 convention), intersection 50: IoU = 50 / 100 = 0.5 exactly
    C, D = [0, 0, 9, 9], [5, 0, 14, 9]         # +1: inter 5 x 10 = 50, union 150: IoU = 1/3.   Without +1 it would be 36 / 126 = 0.2857
    E, F_ = [0, 0, 9, 9], [9, 0, 18, 9]        # share only the column x = 9: +1 convention gives inter 1 x 10 = 10, IoU = 10 / 190
    # chain: P suppresses Q (7 x 10 = 70 / 130 = 0.538 > 0.5); R overlaps Q the same way but P only 40 / 160 = 0.25 -> R survives
    P, Q, R = [0, 0, 9, 9], [3, 0, 12, 9], [6, 0, 15, 9]
    return [
        (f([A + [0.9], B + [0.8]]), 0.5, [0, 1]),               # IoU == threshold: NOT suppressed (strict >)
        (f([A + [0.9], B + [0.8]]), 0.4999, [0]),
        (f([C + [0.9], D + [0.8]]), 0.3, [0]),                  # suppressed only under the +1 convention
        (f([C + [0.9], D + [0.8]]), 1.0 / 3.0 + 1e-3, [0, 1]),
        (f([E + [0.9], F_ + [0.8]]), 0.05, [0]),                # 10 / 190 = 0.0526 > 0.05
        (f([E + [0.9], F_ + [0.8]]), 0.06, [0, 1]),
        (f([P + [0.9], Q + [0.8], R + [0.7]]), 0.5, [0, 2]),    # greedy: Q is gone before it could suppress R
        (f([R + [0.7], Q + [0.8], P + [0.9]]), 0.5, [0, 2]),    # same boxes, input order reversed: indices refer to the input
        (f([P + [0.5], P + [0.6], P + [0.7]]), 0.99, [2]),      # identical boxes: IoU = 1, only the best score stays
    ]


def run_csp_ref_clip(G, tag, device, engine="fused", graph=0, channels_last=False, fold_bn=False, repeats=1):
    """Replays clip ``tag`` of tests/golden/csp_ref_modules.npz -- what the REFERENCE's own mmdet detector (CSPBlockCopy built from
    the C5 config; oracle/gen_golden.py gen_csp_ref_modules) produced -- through this repo's CSP-ResNet50 + CSPBlockCopy manager.
    Returns a dict of worst-case figures over the frames: ``maps`` (head maps, relative to max(1, |golden|max)), ``neck`` (strided
    sample of the packed 768-channel neck output, same scale), ``decode`` (boxes decoded by THIS repo from the FIXTURE's maps vs the
    fixture's boxes, absolute; -1 if the counts differ), ``e2e_match`` (fraction of fixture boxes the end-to-end boxes reproduce
    within 1e-3)."""
    import json

    from blockcopy.core import tensorwrapper as tw
    from blockcopy.core.argparser import default_settings
    from bc_workloads import seeded
    from bc_workloads.bn_fold import fold_batchnorm
    from bc_workloads.csp import CSP, CSPBlockCopy

    cfg = json.loads(bytes(G[f"{tag}_cfg"]).decode())
    tw.set_engine(engine)
    grids = [torch.from_numpy(G[f"{tag}_grid{t}"]) for t in range(cfg["n_frames"])]
    base = CSP()
    base.load_state_dict(seeded.csp_reference_weights(dict(base.state_dict())), strict=True)
    det = CSPBlockCopy(default_settings(block_policy="all", block_size=cfg["block_size"], block_graph=graph), results="device",
                       arch=(base.backbone, base.neck, base.bbox_head))
    det.policy = make_forced_policy(cfg["block_size"], grids * repeats)
    det.eval()
    det = det.to(device)
    if fold_bn:
        det = fold_batchnorm(det)
    if channels_last:
        det = det.to(memory_format=torch.channels_last)
    rec = {}

    sub = cfg.get("subsample")       # full-size clip: strided lattices of the head maps instead of the maps (oracle/gen_golden.py)

    def neck_hook(mod, inp, out):
        o = out[0]
        o = o._plain() if hasattr(o, "_plain") else o
        rec["neck"] = (o[:, ::16, ::4, ::4] if sub is None else o[:, ::64, ::8, ::8]).float().cpu().clone()

    hook = det.neck.register_forward_hook(neck_hook) if not graph else None     # (no host reads inside a graph capture)
    worst = dict(maps=0.0, neck=0.0, decode=0.0, e2e_match=1.0)
    tc = cfg["test_cfg"]
    kw = dict(nms_pre=tc["nms_pre"], score_thr=tc["score_thr"], iou_thr=tc["iou_thr"], max_per_img=tc["max_per_img"])
    try:
        for rep in range(repeats):
            det.reset_temporal()
            for t in range(cfg["n_frames"]):
                x = seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"])).to(device)
                with torch.no_grad():
                    dets, labels = det.simple_test(x)
                for k, m in zip(("cls", "reg", "offset"), det.head_out):
                    m = m.float().cpu()
                    if sub is None:
                        want = torch.from_numpy(G[f"{tag}_{k}{t}"])
                        assert tuple(m.shape) == tuple(want.shape), (k, t, m.shape, want.shape)
                        worst["maps"] = max(worst["maps"], float((m - want).abs().max()) / max(1.0, float(want.abs().max())))
                    else:
                        scale = max(1.0, float(G[f"{tag}_{k}{t}_absmax"]))
                        for o in sub["offsets"]:
                            want = torch.from_numpy(G[f"{tag}_{k}{t}_o{o}"])
                            worst["maps"] = max(worst["maps"], float((m[:, :, o::sub["step"], o::sub["step"]] - want).abs().max()) / scale)
                        worst["maps"] = max(worst["maps"], abs(float(m.abs().max()) - float(G[f"{tag}_{k}{t}_absmax"])) / scale)
                if not graph:    # (a captured body does not run Python hooks on replay)
                    want = torch.from_numpy(G[f"{tag}_neck{t}"])
                    assert rec["neck"].shape == want.shape, (rec["neck"].shape, want.shape)
                    worst["neck"] = max(worst["neck"], float((rec["neck"] - want).abs().max()) / max(1.0, float(want.abs().max())))
                boxes = torch.from_numpy(G[f"{tag}_boxes{t}"])
                if sub is None:
                    # decode + NMS of the FIXTURE's maps: every decision (top-k, threshold, suppression, cap) has the reference's inputs
                    fm = [torch.from_numpy(G[f"{tag}_{k}{t}"]).to(device) for k in ("cls", "reg", "offset")]
                    got, lab = det.bbox_head.get_bboxes(*fm, img_shape=(cfg["H"], cfg["W"]), **kw)
                    if tuple(got.shape) != tuple(boxes.shape) or int(lab.abs().sum()) != 0:
                        worst["decode"] = -1.0
                    elif worst["decode"] >= 0 and boxes.numel():
                        worst["decode"] = max(worst["decode"], float((got.float().cpu() - boxes).abs().max()))
                # end to end: own maps differ by rounding, so a borderline decision may flip -- count the boxes reproduced
                e2e = dets.float().cpu()
                if boxes.shape[0]:
                    d = (boxes[:, None, :] - e2e[None, :, :]).abs().amax(dim=2) if e2e.shape[0] else torch.full((boxes.shape[0], 1), 1e9)
                    worst["e2e_match"] = min(worst["e2e_match"], float((d.amin(dim=1) <= 1e-3).float().mean()))
    finally:
        if hook is not None:
            hook.remove()
        tw.set_engine("fused")
    return worst
