"""Generate tests/golden/*.npz by running the REFERENCE Python (see ref_loader.py) in this container.

Usage (build container only):  python oracle/gen_golden.py

Fixtures are data only: inputs are re-derivable from seeds (bc_workloads.seeded), outputs are what the
reference's own TensorWrapper / BlockCopyModel / SwiftNet code produced on CPU with the oracle standing in
for the four CUDA kernels.  The script also checks the reference-independent properties P1/P2 (SURVEY.md
section 4) that pin the oracle's kernel restatements, and records them in properties.json.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
GOLD = os.path.join(ROOT, "tests", "golden")

import ref_loader  # noqa: E402
from bc_workloads import seeded  # noqa: E402  (pure helpers; does not import our `blockcopy`)

torch.manual_seed(0)
torch.set_num_threads(8)
torch.backends.mkldnn.deterministic = True if hasattr(torch.backends.mkldnn, "deterministic") else None

SETTINGS = dict(block_policy="all", block_num_classes=19, block_optim_lr=1e-4, block_optim_wd=1e-3,
                block_optim_momentum=0, block_target=0.5, block_complexity_weight=5, block_size=128,
                block_train_interval=4, block_cost_momentum=0.9, block_policy_verbose=False)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


# ----------------------------------------------------------------------------- A. index tables
def gen_index_tables(ref):
    rng = np.random.default_rng(1234)
    out = {}
    cases = [(1, 2, 4), (2, 3, 3), (1, 8, 16), (1, 1, 1), (2, 1, 5), (1, 32, 64)]
    meta = []
    for ci, (N, GH, GW) in enumerate(cases):
        total = N * GH * GW
        fracs = [1.0, 0.5, 0.25, "one", 0.5, "allbutone", 0.5]
        prev = None
        for fi, fr in enumerate(fracs):
            if fr == "one":
                g = np.zeros(total, bool); g[rng.integers(total)] = True
            elif fr == "allbutone":
                g = np.ones(total, bool); g[rng.integers(total)] = False
            elif fr == 1.0:
                g = np.ones(total, bool)
            else:
                g = rng.random(total) < fr
                if not g.any():
                    g[0] = True
            grid = torch.from_numpy(g.reshape(N, 1, GH, GW))
            bf = ref.tw.BlockFeatures(device="cpu")
            bf._process_grid(grid, prev)
            k = f"c{ci}_f{fi}"
            out[k + "_grid"] = g.reshape(N, 1, GH, GW)
            out[k + "_grid_idx"] = bf._grid_idx.numpy()
            out[k + "_mapping_exec"] = bf._mapping_exec.numpy()
            out[k + "_transfer_idx"] = (bf._transfer_idx.numpy() if bf._transfer_idx is not None else np.zeros(0, np.int32))
            prev = bf
        meta.append(dict(case=ci, N=N, GH=GH, GW=GW, frames=len(fracs)))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, "index_tables.npz"), **out)
    print("index_tables.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- B. op-level call log
class TinyNet(torch.nn.Module):
    """conv k3p1 -> relu -> maxpool k3s2p1 -> conv k5p2 -> relu -> conv k3 s2 p1 ; exercises p in {1,2} and tile 8->4->2."""

    def __init__(self):
        super().__init__()
        self.c1 = torch.nn.Conv2d(3, 4, 3, padding=1)
        self.c2 = torch.nn.Conv2d(4, 5, 5, padding=2)
        self.c3 = torch.nn.Conv2d(5, 6, 3, stride=2, padding=1)

    def forward(self, x):
        x = torch.relu(self.c1(x))
        x = torch.nn.functional.max_pool2d(x, 3, 2, 1)
        x = torch.relu(self.c2(x))
        return self.c3(x)


def tiny_net_weights(net):
    net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()))
    return net


def tiny_grids():
    N, GH, GW = 2, 2, 3
    gs = [torch.ones(N, 1, GH, GW, dtype=torch.bool)]
    for s, n in ((11, 6), (12, 3), (13, 1), (14, 11)):
        gs.append(seeded.fixed_fraction_grid(s, N, GH, GW, n))
    return gs


def gen_ops(ref):
    net = tiny_net_weights(TinyNet()).eval()
    st = dict(SETTINGS, block_size=8)
    model = ref.bc.BlockCopyModel(net, st)
    grids = tiny_grids()
    model.policy = ref_loader.make_forced_policy(ref, 8, grids)
    model.reset_temporal()
    ref_loader.CALL_LOG = []
    outs = []
    with torch.no_grad():
        for t in range(len(grids)):
            x = seeded.synthetic_frame(100 + t, (2, 3, 16, 24))
            outs.append(model(x).clone())
    log, ref_loader.CALL_LOG = ref_loader.CALL_LOG, None
    out = {}
    names = []
    for i, (name, kw) in enumerate(log):
        names.append(name)
        for k, v in kw.items():
            out[f"k{i}_{k}"] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    out["names"] = np.frombuffer(json.dumps(names).encode(), dtype=np.uint8)
    for t, o in enumerate(outs):
        out[f"net_out{t}"] = o.numpy()
    np.savez_compressed(os.path.join(GOLD, "ops_tinynet.npz"), **out)
    print("ops_tinynet.npz", len(log), "kernel calls:", {n: names.count(n) for n in set(names)})


# ----------------------------------------------------------------------------- C. SwiftNet end to end
def build_ref_swiftnet(ref, backbone, block_size, grids):
    with quiet():
        bb = getattr(ref.resnet, backbone)(pretrained=False)
        model = ref.swiftnet.SwiftNet(backbone=bb, num_classes=19, num_features=128, use_spp=True)
        model.load_state_dict(seeded.name_seeded_state_dict(model.state_dict()), strict=True)
        model.eval()
        dense_model = None
        wrapped = ref.bc.BlockCopyModel(model, dict(SETTINGS, block_size=block_size))
        wrapped.policy = ref_loader.make_forced_policy(ref, block_size, grids)
        wrapped = ref.bn_fusion.fuse_bn_recursively(wrapped)
    return wrapped, dense_model


def scenario_grids(N, GH, GW, seed):
    total = N * GH * GW
    gs = [torch.ones(N, 1, GH, GW, dtype=torch.bool),
          seeded.fixed_fraction_grid(seed + 1, N, GH, GW, total // 2),
          seeded.fixed_fraction_grid(seed + 2, N, GH, GW, 1),
          seeded.fixed_fraction_grid(seed + 3, N, GH, GW, total - 1),
          torch.zeros(N, 1, GH, GW, dtype=torch.bool),
          seeded.fixed_fraction_grid(seed + 5, N, GH, GW, total // 4)]
    return gs


def gen_swiftnet(ref, tag, backbone, N, H, W, bs, n_frames, seed, store_frame_state, subsample=None, grids=None):
    """``subsample=(step, offsets)``: store ``logits[..., o::step, o::step]`` for every offset instead of the full map (full-size
    clips: the lattices are chosen so that both border pixels of every logits tile are among the samples)."""
    grids = scenario_grids(N, H // bs, W // bs, seed)[:n_frames] if grids is None else grids
    model, _ = build_ref_swiftnet(ref, backbone, bs, grids)
    model.reset_temporal()
    cfg = dict(backbone=backbone, N=N, H=H, W=W, block_size=bs, n_frames=n_frames, frame_seed0=seed * 1000)
    if subsample is not None:
        cfg["subsample"] = dict(step=subsample[0], offsets=list(subsample[1]))
    out = {"cfg": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)}
    import warnings
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(n_frames):
            x = seeded.synthetic_frame(seed * 1000 + t, (N, 3, H, W))
            y = model(x)
            out[f"grid{t}"] = grids[t].numpy()
            if subsample is None:
                out[f"logits{t}"] = y.detach().numpy().copy()
            else:
                for o in subsample[1]:
                    out[f"logits{t}_o{o}"] = y.detach()[:, :, o::subsample[0], o::subsample[0]].numpy().copy()
                out[f"logits{t}_absmax"] = np.float32(y.detach().abs().max())
                out[f"logits{t}_chansum"] = y.detach().double().sum(dim=(0, 2, 3)).numpy()
            if store_frame_state:
                out[f"frame_state{t}"] = model.policy_meta["frame_state"].detach().numpy().copy()
    np.savez_compressed(os.path.join(GOLD, f"swiftnet_{tag}.npz"), **out)
    print(f"swiftnet_{tag}.npz", {k: v.shape for k, v in out.items() if k.startswith("logits")})


# ----------------------------------------------------------------------------- D. properties P1/P2/P3
def check_properties(ref):
    import warnings
    props = {}
    N, H, W, bs = 1, 256, 512, 64
    GH, GW = H // bs, W // bs
    # dense reference model (no wrapper), same weights, BN folded
    with quiet():
        bb = ref.resnet.resnet18(pretrained=False)
        dense = ref.swiftnet.SwiftNet(backbone=bb, num_classes=19, num_features=128, use_spp=True)
        dense.load_state_dict(seeded.name_seeded_state_dict(dense.state_dict()), strict=True)
        dense.eval()
        dense = ref.bn_fusion.fuse_bn_recursively(dense)
    x0 = seeded.synthetic_frame(7, (N, 3, H, W))
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dense_feats = dense.forward_down(x0)
        dense_logits = dense(x0)
        # P1: all-active block path == dense encoder
        grids = [torch.ones(N, 1, GH, GW, dtype=torch.bool)]
        blk, _ = build_ref_swiftnet(ref, "resnet18", bs, grids)
        blk.reset_temporal()
        xw = x0.as_subclass(ref.tw.TensorWrapper)
        feats_obj = xw.process_temporal_features(None)
        xb = xw.to_blocks(grids[0])
        block_feats = blk.base_model.forward_down(xb)
        p1 = []
        for d, b in zip(dense_feats, block_feats):
            full = b.combine().to_tensor()
            p1.append(float((full - d).abs().max() / d.abs().max()))
        props["P1_encoder_rel_maxabs_per_level"] = p1
        assert max(p1) < 1e-5, p1  # levels 0-1 exact; deeper levels differ only by conv summation order
        # P3: decoder differs from dense even at 100 % (no-halo per-tile bilinear upsample)
        blk2, _ = build_ref_swiftnet(ref, "resnet18", bs, grids)
        blk2.reset_temporal()
        l_all = blk2(x0)
        props["P3_logits_block_vs_dense_maxabs"] = float((l_all - dense_logits).abs().max())
        props["logits_absmax"] = float(dense_logits.abs().max())
        # P2: static clip, arbitrary masks -> logits identical to frame 0
        gs = scenario_grids(N, GH, GW, 77)
        blk3, _ = build_ref_swiftnet(ref, "resnet18", bs, gs)
        blk3.reset_temporal()
        l0 = blk3(x0).clone()
        p2 = []
        for t in range(1, len(gs)):
            lt = blk3(x0)
            p2.append(float((lt - l0).abs().max() / l0.abs().max()))
        props["P2_static_clip_rel_maxabs_vs_frame0"] = p2
        assert max(p2) < 1e-5, p2
    with open(os.path.join(GOLD, "properties.json"), "w") as f:
        json.dump(props, f, indent=1)
    print("properties:", props)


def gen_keys(ref):
    """Parameter names/shapes of the reference SwiftNet (pins state_dict compatibility of bc_workloads.swiftnet)."""
    out = {}
    for bb in ("resnet18", "resnet50"):
        with quiet():
            m = ref.swiftnet.SwiftNet(backbone=getattr(ref.resnet, bb)(pretrained=False), num_classes=19, num_features=128, use_spp=True)
        out[bb] = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(GOLD, "swiftnet_keys.json"), "w") as f:
        json.dump(out, f)
    print("swiftnet_keys.json", {k: len(v) for k, v in out.items()})


# ----------------------------------------------------------------------------- E. one online-RL policy run (config C3)
def gen_rl_c3(ref):
    """BASELINE config C3 at its FULL size through the reference: SwiftNet-RN18 1x3x1024x2048, block 128, rl_semseg at target 0.3,
    train_interval 3 (bench.py --config C3), 5 frames; logits as two 8-strided lattices."""
    gen_rl(ref, name="rl_semseg_c3.npz", N=1, H=1024, W=2048, bs=128, target=0.3, interval=3, n_frames=5, frame_seed0=9100, subsample=(8, (0, 7)))


def gen_rl(ref, name="rl_semseg_run.npz", N=1, H=128, W=256, bs=32, target=0.4, interval=2, n_frames=4, frame_seed0=4242, subsample=None):
    """``n_frames`` frames of SwiftNet-RN18 under the reference's rl_semseg policy (PolicyTrainRL + PolicyNet +
    InformationGainSemSeg + RMSprop).  Seeds: torch/random = 0 right before the first frame; the
    policy net gets name-seeded weights so the fixture does not depend on constructor RNG order."""
    import random
    import warnings

    with quiet():
        bb = ref.resnet.resnet18(pretrained=False)
        model = ref.swiftnet.SwiftNet(backbone=bb, num_classes=19, num_features=128, use_spp=True)
        model.load_state_dict(seeded.name_seeded_state_dict(model.state_dict()), strict=True)
        model.eval()
        st = dict(SETTINGS, block_policy="rl_semseg", block_size=bs, block_target=target, block_train_interval=interval)
        wrapped = ref.bc.BlockCopyModel(model, st)
        wrapped.policy.net.load_state_dict(seeded.name_seeded_state_dict(wrapped.policy.net.state_dict()))
        wrapped = ref.bn_fusion.fuse_bn_recursively(wrapped)
    assert wrapped.policy.net.training
    torch.manual_seed(0)
    random.seed(0)
    wrapped.reset_temporal()
    cfg = dict(N=N, H=H, W=W, block_size=bs, n_frames=n_frames, block_target=target, train_interval=interval, frame_seed0=frame_seed0)
    if subsample is not None:
        cfg["subsample"] = dict(step=subsample[0], offsets=list(subsample[1]))
    out = {"cfg": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)}
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(n_frames):
            x = seeded.synthetic_frame(frame_seed0 + t, (N, 3, H, W))
            y = wrapped(x)
            pm = wrapped.policy_meta
            out[f"grid{t}"] = pm["grid"].numpy().copy()
            if subsample is None:
                out[f"logits{t}"] = y.detach().numpy().copy()
            else:
                for o in subsample[1]:
                    out[f"logits{t}_o{o}"] = y.detach()[:, :, o::subsample[0], o::subsample[0]].numpy().copy()
                out[f"logits{t}_absmax"] = np.float32(y.detach().abs().max())
                out[f"logits{t}_chansum"] = y.detach().double().sum(dim=(0, 2, 3)).numpy()
            if t > 0:
                out[f"grid_probs{t}"] = pm["grid_probs"].detach().numpy().copy()
            if "information_gain" in pm and wrapped.clip_length % interval == 0:
                out[f"information_gain{t}"] = pm["information_gain"].detach().numpy().copy()
            out[f"running_cost{t}"] = np.float64(wrapped.policy.running_cost)
    sd = wrapped.policy.net.state_dict()
    out["policy_abs_sum_after"] = np.float64(sum(float(v.double().abs().sum()) for k, v in sd.items() if v.dtype.is_floating_point))
    out["policy_conv1_after"] = sd["backbone.conv1.weight"].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, name), **out)
    print(name, "exec per frame:", [int(out[f"grid{t}"].sum()) for t in range(n_frames)], "running_cost", float(out[f"running_cost{n_frames - 1}"]))


# ----------------------------------------------------------------------------- F. detector op classes (Pedestron CSP path)
def _detector_fixture(ref, name, modules, cfg, grids, extra=None):
    """Drive (backbone, neck, head) through the REFERENCE TensorWrapper with forced grids; store head maps + frame_state."""
    import warnings

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tinycsp

    for m in modules:
        m.eval()
    frames = [seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"])) for t in range(cfg["n_frames"])]
    ref_loader.CALL_LOG = []
    with warnings.catch_warnings(), quiet():
        warnings.simplefilter("ignore")
        res = tinycsp.run_detector_clip(modules, frames, grids)
    log, ref_loader.CALL_LOG = ref_loader.CALL_LOG, None
    names = [n for n, _ in log]
    out = {"cfg": np.frombuffer(json.dumps(dict(cfg, **(extra or {}))).encode(), dtype=np.uint8),
           "kernel_calls": np.frombuffer(json.dumps({n: names.count(n) for n in sorted(set(names))}).encode(), dtype=np.uint8)}
    for t, (maps, fs) in enumerate(res):
        out[f"grid{t}"] = grids[t].numpy()
        for k, m in zip(("cls", "reg", "offset"), maps):
            out[f"{k}{t}"] = m.numpy().copy()
        if t < 2:
            out[f"frame_state{t}"] = fs.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, name), **out)
    print(name, {n: names.count(n) for n in sorted(set(names))}, "cls range", float(res[0][0][0].min()), float(res[0][0][0].max()))


def gen_tinycsp(ref):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tinycsp

    modules = tinycsp.seed_weights(tinycsp.tinycsp_arch())
    _detector_fixture(ref, "tinycsp.npz", modules, tinycsp.CFG, tinycsp.tinycsp_grids())


def gen_csp_r50(ref):
    """The repo's own restatement of the CSP-ResNet50 detector (bc_workloads/csp.py: backbone / neck / head modules,
    pure torch) driven through the REFERENCE TensorWrapper at 128x256, block 32: pins the C5 op sequence (dilated stage,
    transposed-conv neck, GroupNorm head, three to_tensor branches) against the reference's routing."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tinycsp
    from bc_workloads import csp   # imports `blockcopy` = the reference here (only its public names are used by the modules)

    assert csp.blockcopy.__file__.startswith(ref_loader.REF_ROOT)
    det = csp.CSP()
    core = {k: v for k, v in det.state_dict().items()}
    det.load_state_dict(seeded.name_seeded_state_dict(core), strict=True)
    cfg = dict(N=1, H=128, W=256, block_size=32, n_frames=4, frame_seed0=52000, grid_seed=83)
    _detector_fixture(ref, "csp_r50.npz", (det.backbone, det.neck, det.bbox_head), cfg, tinycsp.tinycsp_grids(cfg),
                      extra=dict(weights="name-seeded over CSP().state_dict() keys, BN not folded (Pedestron does not fold)"))


csp_ref_weights = seeded.csp_reference_weights
CSP_REF_OVERRIDES = seeded.CSP_REF_OVERRIDES


def gen_csp_ref_modules(ref):
    """The reference's OWN detector -- mmdet CSPBlockCopy / CSP (detectors/csp_blockcopy.py:46-95), ResNet (backbones/resnet.py),
    CSPNeck (necks/csp_neck.py:68-83), CSPHead.forward_single (anchor_heads/csp_head.py:130-152) and get_bboxes_single (:230-284:
    sigmoid / exp decode, top-1000, csp_height2bbox core/bbox/transforms.py:182-212, multiclass_nms) -- built from the C5 config
    (csp_r50_clip_blockcopy_030.py) and run through the reference TensorWrapper with forced grids.  See ref_loader.load_reference_csp
    for what is a stand-in (mmcv initialisers, the compiled NMS extension -> oracle restatement of nms_kernel.cu).
    Clip a: 128x256, block 32 (tiles down to 2x2 with dilation 2: the reference's padding == tile size regime), 4 frames + the dense
    detector on frame 0; clip b: 256x512, block 128 (the tile sizes of C5 itself: 128 -> 64 -> 32 -> 16 -> 8), 3 frames."""
    import warnings

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tinycsp

    ped = ref_loader.load_reference_csp(ref)
    out = {}
    clips = {"a": dict(N=1, H=128, W=256, block_size=32, n_frames=4, frame_seed0=77000, grid_seed=29),
             "b": dict(N=1, H=256, W=512, block_size=128, n_frames=3, frame_seed0=78000, grid_seed=31),
             # BASELINE config C5 at its FULL size: 1024x2048, block 128 (grid 8x16), frame 0 all-active, then 38 of 128 tiles (30 %);
             # the stride-4 head maps are stored as two 8-strided lattices (offsets 0 and 7) + absolute maximum
             "c": dict(N=1, H=1024, W=2048, block_size=128, n_frames=3, frame_seed0=79000, grid_seed=37, subsample=dict(step=8, offsets=[0, 7]))}
    only = set(os.environ.get("CSP_REF_CLIPS", "a,b,c").split(","))
    if only != {"a", "b", "c"}:      # regenerate a subset: keep the other clips of the existing fixture
        out.update({k: v for k, v in np.load(os.path.join(GOLD, "csp_ref_modules.npz")).items()})
    for tag, cfg in clips.items():
        if tag not in only:
            continue
        model_cfg, test_cfg = ref_loader.csp_r50_config(cfg["block_size"])
        with quiet():
            det = ped.det.CSPBlockCopy(train_cfg=None, test_cfg=test_cfg, **model_cfg)
        sd = det.state_dict()
        if tag == "a":
            out["state_dict_keys"] = np.frombuffer(json.dumps({k: list(v.shape) for k, v in sd.items()}).encode(), dtype=np.uint8)
        det.load_state_dict(csp_ref_weights(sd), strict=True)
        det.eval()
        grids = tinycsp.tinycsp_grids(cfg)
        if tag == "c":
            GH, GW = cfg["H"] // cfg["block_size"], cfg["W"] // cfg["block_size"]
            grids = [torch.ones(1, 1, GH, GW, dtype=torch.bool)] + [seeded.fixed_fraction_grid(cfg["grid_seed"] + t, 1, GH, GW, 38) for t in (1, 2)]
        det.policy = ref_loader.make_forced_policy(ref, cfg["block_size"], grids)
        rec = {}
        h1 = det.neck.register_forward_hook(lambda m, i, o: rec.__setitem__("neck", o[0].as_subclass(torch.Tensor).detach().clone()))
        h2 = det.bbox_head.register_forward_hook(lambda m, i, o: rec.__setitem__("maps", [x[0].detach().clone() for x in o]))
        meta = [dict(img_shape=(cfg["H"], cfg["W"], 3), scale_factor=1.0)]
        det.reset_temporal()
        kept = []
        for t in range(cfg["n_frames"]):
            img = seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"]))
            with torch.no_grad(), warnings.catch_warnings(), quiet():
                warnings.simplefilter("ignore")
                res = det.simple_test(img, meta, rescale=False)
            assert isinstance(res, list) and len(res) == 1 and res[0].shape[1] == 5
            out[f"{tag}_grid{t}"] = grids[t].numpy()
            sub = cfg.get("subsample")
            for k, m in zip(("cls", "reg", "offset"), rec["maps"]):
                if sub is None:
                    out[f"{tag}_{k}{t}"] = m.numpy().copy()
                else:
                    for o in sub["offsets"]:
                        out[f"{tag}_{k}{t}_o{o}"] = m[:, :, o::sub["step"], o::sub["step"]].numpy().copy()
                    out[f"{tag}_{k}{t}_absmax"] = np.float32(m.abs().max())
            out[f"{tag}_neck{t}"] = rec["neck"][:, ::16, ::4, ::4].numpy().copy() if sub is None else rec["neck"][:, ::64, ::8, ::8].numpy().copy()      # packed tiles, strided sample
            out[f"{tag}_boxes{t}"] = res[0].copy()                                       # bbox2result(...)[class 0]: (k, 5) after NMS, top max_per_img
            kept.append(res[0].shape[0])
            scores = rec["maps"][0].sigmoid().reshape(-1)
            print(f"csp_ref_modules {tag} frame {t}: n_exec {int(grids[t].sum())}, scores > thr among top-1000: {int((scores.topk(min(1000, scores.numel()))[0] > 0.1).sum())}, kept {kept[-1]}")
        h1.remove(), h2.remove()
        if tag == "a":
            # the dense detector (detectors/csp.py + single_stage.py:61-70) on frame 0: pins the model without the block path
            with quiet():
                dense = ped.csp.CSP(model_cfg["backbone"], model_cfg["neck"], model_cfg["bbox_head"], train_cfg=None, test_cfg=test_cfg, pretrained=None)
            dense.load_state_dict(csp_ref_weights(dense.state_dict()), strict=True)
            dense.eval()
            h2 = dense.bbox_head.register_forward_hook(lambda m, i, o: rec.__setitem__("maps", [x[0].detach().clone() for x in o]))
            img = seeded.synthetic_frame(cfg["frame_seed0"], (cfg["N"], 3, cfg["H"], cfg["W"]))
            with torch.no_grad(), quiet():
                res = dense.simple_test(img, meta, rescale=False)
            h2.remove()
            for k, m in zip(("cls", "reg", "offset"), rec["maps"]):
                out[f"dense_{k}"] = m.numpy().copy()
            out["dense_boxes"] = res[0].copy()
        out[f"{tag}_cfg"] = np.frombuffer(json.dumps(dict(cfg, weights="csp_ref_weights: name-seeded over the reference state_dict keys, prediction convs rescaled by overrides, "
                                                          "scales = 1; BN not folded (Pedestron does not fold)", overrides=CSP_REF_OVERRIDES,
                                                          test_cfg=dict(nms_pre=1000, score_thr=0.1, iou_thr=0.5, max_per_img=100))).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, "csp_ref_modules.npz"), **out)
    print("csp_ref_modules.npz", os.path.getsize(os.path.join(GOLD, "csp_ref_modules.npz")), "bytes")


# ----------------------------------------------------------------------------- G. I/O format, quality metrics, GMACs counter
def gen_io_metrics(ref):
    """Fixtures for the components either side of the hot path, from the reference's own
    CityscapesVid / StreamSegMetrics / flopscounter run in this container (cv2 is only imported, never called, by these
    modules: a bare stub module satisfies the import; torchvision-based transforms are not used)."""
    import importlib.util
    import tempfile
    import types
    import warnings

    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    from lib.datasets.cityscapes_vid import CityscapesVid
    from lib.utils.metrics import StreamSegMetrics
    from bc_workloads import cityscapes as cs

    out = {}
    # (1) clip layout + label encoding: a synthetic tree written by bc_workloads.cityscapes.write_synthetic_tree (data only)
    cfg = dict(split="val", cities=["aachen", "bonn"], clips_per_city=2, clip_length=4, size=[24, 48], seed=5)
    with tempfile.TemporaryDirectory() as root:
        cs.write_synthetic_tree(root, cfg["split"], cfg["cities"], cfg["clips_per_city"], cfg["clip_length"], tuple(cfg["size"]), cfg["seed"])
        plain = lambda img, lbl: (np.array(img, dtype=np.uint8), lbl)     # noqa: E731  (PIL -> array, label untouched)
        with quiet():
            ds = CityscapesVid(root, split=cfg["split"], transform=plain, clip_length=cfg["clip_length"], has_labels=True)
        order = sorted(range(len(ds)), key=lambda i: ds.relative_dirs[i])    # os.listdir order is filesystem-dependent
        rels = []
        for k, i in enumerate(order):
            vid, target, meta = ds[i]
            out[f"clip{k}_frames"] = np.stack(vid)
            out[f"clip{k}_target"] = np.asarray(target)
            rels.append(meta["relpath"])
        out["io_cfg"] = np.frombuffer(json.dumps(dict(cfg, relpaths=rels, mean=list(CityscapesVid.mean), std=list(CityscapesVid.std),
                                                      fine_classes=list(CityscapesVid.fine_classes))).encode(), dtype=np.uint8)
        out["encode_all_ids"] = CityscapesVid.encode_target(np.arange(-1, 34))
        out["decode_all_train_ids"] = CityscapesVid.decode_target(np.array(list(range(19)) + [255]))
        out["encode_test_all"] = CityscapesVid.encode_target_test(np.arange(19))

    # (2) StreamSegMetrics: three updates of seeded labels / predictions (255 = ignore present)
    rng = np.random.default_rng(77)
    m = StreamSegMetrics(19, classes=CityscapesVid.fine_classes)
    res = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for u in range(3):
            lt = rng.integers(0, 19, (2, 16, 32))
            lt[rng.random(lt.shape) < 0.1] = 255
            if u == 0:
                lt[lt == 16] = 255           # a class absent from the first update: NaN handling
            lp = np.where(rng.random(lt.shape) < 0.6, np.where(lt == 255, 0, lt), rng.integers(0, 19, lt.shape))
            m.update(lt, lp)
            out[f"m_lt{u}"], out[f"m_lp{u}"] = lt, lp
            r = m.get_results()
            res.append({k: (float(v) if k != "Class IoU" else {kk: float(vv) for kk, vv in v.items()}) for k, v in r.items()})
    out["m_confusion"] = m.confusion_matrix
    out["m_ious"], out["m_ious_sum"] = np.array(m.ious), np.array(m.ious_sum)
    out["m_accs"], out["m_accs_sum"] = np.array(m.accs), np.array(m.accs_sum)
    out["m_results"] = np.frombuffer(json.dumps(res).encode(), dtype=np.uint8)

    # (3) GMACs: the reference's counter on the reference SwiftNet-RN18, dense and under block execution (packed batch = executed tiles)
    spec = importlib.util.spec_from_file_location("ref_flopscounter", os.path.join(ref_loader.REF_ROOT, "Pedestron", "tools", "flopscounter.py"))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    N, H, W, bs = 1, 128, 256, 32
    grids = scenario_grids(N, H // bs, W // bs, 21)[:4]
    frames = [seeded.synthetic_frame(6000 + t, (N, 3, H, W)) for t in range(4)]
    g = {}
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with quiet():
            bb = ref.resnet.resnet18(pretrained=False)
            dense = ref.swiftnet.SwiftNet(backbone=bb, num_classes=19, num_features=128, use_spp=True).eval()
        fc.add_flops_counting_methods(dense)
        dense.start_flops_count()
        dense(frames[0])
        dense(frames[1])
        g["dense_avg"], g["dense_frames"] = [float(x) for x in dense.compute_average_flops_cost()]
        dense.stop_flops_count()
        blk, _ = build_ref_swiftnet(ref, "resnet18", bs, grids)     # BN folded, forced grids
        fc.add_flops_counting_methods(blk)
        blk.start_flops_count()
        blk.reset_temporal()
        per_frame = []
        for t in range(4):
            blk(frames[t])
            tot, _, n = blk.compute_total_flops_cost()
            per_frame.append(float(tot))
        g["block_avg"], g["block_frames"] = [float(x) for x in blk.compute_average_flops_cost()]
        g["block_total_after_frame"] = per_frame
        blk.stop_flops_count()
    g.update(N=N, H=H, W=W, block_size=bs, grid_seed=21, frame_seed0=6000, exec=[int(x.sum()) for x in grids])
    out["gmacs"] = np.frombuffer(json.dumps(g).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, "io_metrics.npz"), **out)
    print("io_metrics.npz", rels, "mIoU", res[-1]["Mean IoU"], "GMACs dense", g["dense_avg"] / 1e9, "block avg", g["block_avg"] / 1e9, g["exec"])


def gen_swiftnet_rn18_c(ref):
    """BASELINE config C2's TILE GEOMETRY (block 128: the stem on 128-px tiles, then 32x32 ... 4x4 tiles, the SPP on a
    2x4 grid) at a size the CPU reference finishes in seconds: 256x512, grid 2x4, masks all / half / one / all-but-one /
    none / quarter.  Build order as semantic_segmentation/test_swiftnet.py:104-115."""
    gen_swiftnet(ref, "rn18_c", "resnet18", 1, 256, 512, 128, 6, 6, False)


def gen_swiftnet_rn18_c2(ref):
    """BASELINE config C2 at its FULL size through the reference: SwiftNet-RN18, 1x3x1024x2048, block 128 (grid 8x16), masks all /
    64 of 128 / one / all-but-one.  The 10 MB logits map of a frame is stored as two 8-strided lattices (offsets 0 and 7: rows /
    columns 0 and 31 of every 32x32 logits tile are sampled) plus its absolute maximum and per-class sums."""
    gen_swiftnet(ref, "rn18_c2", "resnet18", 1, 1024, 2048, 128, 4, 8, False, subsample=(8, (0, 7)))


def gen_swiftnet_rn18_c1(ref):
    """BASELINE config C1 at its stated shape through the reference: SwiftNet-RN18, 4 frames torch.randn(1,3,512,1024) with seeds 0..3,
    block 128 (grid 4x8), every tile executed on every frame.  Logits stored as two 8-strided lattices (see rn18_c2)."""
    grids = [torch.ones(1, 1, 4, 8, dtype=torch.bool) for _ in range(4)]
    gen_swiftnet(ref, "rn18_c1", "resnet18", 1, 512, 1024, 128, 4, 0, False, subsample=(8, (0, 7)), grids=grids)


def gen_swiftnet_rn50_c4(ref):
    """BASELINE config C4 at its FULL size through the reference: SwiftNet-RN50, 1x3x2048x4096, block 64 (grid 32x64 = 2048 tiles:
    tiles 64 -> 32 -> 16 -> 8 -> 4 -> 2, the regime where packed element offsets approach 2^31), frame 0 all-active, frames 1-2 with
    512 of 2048 tiles (25 %).  The 40 MB logits map of a frame is stored as two 16-strided lattices (offsets 0 and 15: rows / columns
    0 and 15 of every 16x16 logits tile) plus its absolute maximum and per-class sums."""
    N, GH, GW = 1, 32, 64
    grids = [torch.ones(N, 1, GH, GW, dtype=torch.bool), seeded.fixed_fraction_grid(4001, N, GH, GW, 512), seeded.fixed_fraction_grid(4002, N, GH, GW, 512)]
    gen_swiftnet(ref, "rn50_c4", "resnet50", 1, 2048, 4096, 64, 3, 12, False, subsample=(16, (0, 15)), grids=grids)


def gen_detgain(ref):
    """The detector-side reward functions of the REFERENCE (blockcopy/blockcopy/policy/information_gain.py:43-108:
    InformationGainObjectDetection.get_output_repr / .forward = build_instance_mask / build_instance_mask_iou_gain) on seeded
    detections.  ``build_instance_mask_iou_gain`` allocates its mask with a hard-coded ``device='cuda'`` (:70); the LOADER (not the
    reference) maps that request to the CPU for the duration of the call.  Boxes are at least 4 pixels wide and high so that none
    collapses at SUBSAMPLE = 2 (the reference asserts there, :137-140).  One class, as in the shipped pedestrian configs (the
    reference's per-class loop rebinds its own arguments, :76-77, so it only runs with one)."""
    import blockcopy.policy.information_gain as ref_ig

    assert ref_ig.__file__.startswith(ref_loader.REF_ROOT)
    rng = np.random.default_rng(17)
    H, W = 96, 160

    def dets(n):
        x1, y1 = rng.integers(0, W - 12, n), rng.integers(0, H - 12, n)
        w, h = rng.integers(4, 48, n), rng.integers(4, 64, n)
        return np.stack([x1, y1, np.minimum(x1 + w, W - 1), np.minimum(y1 + h, H - 1), rng.random(n) * 0.9 + 0.1], 1).astype(np.float32)

    a = dets(12)
    shifted = a.copy(); shifted[:, [0, 2]] += 2
    cases = [(dets(30), dets(25)), (dets(1), dets(0)), (dets(0), dets(7)), (dets(0), dets(0)), (dets(60), dets(75)),
             (a, a.copy()), (a, shifted), (np.repeat(a[:3], 3, 0), a[:3])]
    real_zeros = torch.zeros

    def zeros_on_cpu(*args, **kw):
        if str(kw.get("device", "cpu")).startswith("cuda"):
            kw["device"] = "cpu"
        return real_zeros(*args, **kw)

    ig = ref_ig.InformationGainObjectDetection(num_classes=1)
    out = {"cfg": np.frombuffer(json.dumps(dict(H=H, W=W, n_cases=len(cases), num_classes=1)).encode(), dtype=np.uint8)}
    frame = torch.zeros(1, 3, H, W)
    for k, (cur, prev) in enumerate(cases):
        pm = {"inputs": frame, "outputs": [[cur]], "outputs_prev": [[prev]]}
        torch.zeros = zeros_on_cpu
        try:
            gain = ig(pm)
            rep = ig.get_output_repr(pm)
        finally:
            torch.zeros = real_zeros
        out[f"cur{k}"], out[f"prev{k}"] = cur, prev
        out[f"gain{k}"], out[f"repr{k}"] = gain.numpy().copy(), rep.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "detgain.npz"), **out)
    print("detgain.npz", len(cases), "cases; gain range", float(min(out[f"gain{k}"].min() for k in range(len(cases)))),
          float(max(out[f"gain{k}"].max() for k in range(len(cases)))))


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = ref_loader.load_reference()
    only = set(sys.argv[1:])
    if only:   # regenerate selected fixtures only:  python oracle/gen_golden.py tinycsp csp_r50
        for name in only:
            globals()[f"gen_{name}"](ref)
        return
    gen_tinycsp(ref)
    gen_csp_r50(ref)
    gen_csp_ref_modules(ref)
    gen_keys(ref)
    gen_rl(ref)
    gen_index_tables(ref)
    gen_ops(ref)
    check_properties(ref)
    gen_swiftnet(ref, "rn18_a", "resnet18", 1, 128, 256, 32, 6, 3, True)
    gen_swiftnet(ref, "rn18_b", "resnet18", 1, 256, 512, 64, 4, 5, False)
    gen_swiftnet(ref, "rn18_n2", "resnet18", 2, 128, 128, 32, 4, 9, False)
    gen_swiftnet(ref, "rn50_a", "resnet50", 1, 128, 256, 32, 3, 4, False)
    gen_swiftnet_rn18_c(ref)
    gen_swiftnet_rn18_c2(ref)
    gen_swiftnet_rn18_c1(ref)
    gen_swiftnet_rn50_c4(ref)
    gen_rl_c3(ref)
    gen_detgain(ref)


if __name__ == "__main__":
    main()
