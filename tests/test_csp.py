"""CSP detector workload (BASELINE config C5) on CPU with the oracle-backed checker: mmcv/mmdet cannot be installed
here, so the model is pinned by reference-independent properties instead of reference outputs."""
import json
import os

import numpy as np
import pytest
import torch


def _tiny_inputs(seed=0):
    from bc_workloads import seeded
    return seeded.synthetic_frame(seed, (1, 3, 128, 256))


def test_parameter_names_follow_mmdet(oracle_backend):
    from bc_workloads.csp import CSP

    keys = list(CSP().state_dict())
    for k in ("backbone.conv1.weight", "backbone.layer1.0.downsample.0.weight", "backbone.layer4.2.bn3.running_var",
              "neck.p3.weight", "neck.p3.bias", "neck.p5_l2.weight", "bbox_head.cls_convs.0.conv.weight",
              "bbox_head.reg_convs.0.gn.bias", "bbox_head.offset_convs.0.gn.weight", "bbox_head.csp_cls.bias",
              "bbox_head.csp_offset.weight", "bbox_head.reg_scales.0.scale", "bbox_head.offset_scales.0.scale"):
        assert k in keys, k
    m = CSP()
    assert m.backbone.layer4[0].conv2.dilation == (2, 2) and m.backbone.layer4[0].conv2.padding == (2, 2)
    assert m.backbone.layer4[0].conv2.stride == (1, 1) and m.backbone.layer3[0].conv2.stride == (2, 2)


@pytest.mark.parametrize("graph", [0, 1])
def test_all_active_equals_dense_and_static_clip_is_invariant(oracle_backend, graph):
    """P1: with every tile executed the packed BACKBONE (incl. the dilation-2 stage, halo width 2) equals the dense one
    (the neck's transposed convs run per tile without halo -- reference quirk -- so equality stops there).
    P2: feeding the same frame with arbitrary masks leaves the neck's 768-channel map unchanged; the head maps are only
    invariant when every tile is executed, because GroupNorm statistics run over the executed tiles (reference quirk,
    core/tensorwrapper.py:600-633)."""
    import blockcopy
    from bc_workloads.csp import build_csp

    x = _tiny_inputs()
    dense = build_csp(block_policy="static", device="cpu")
    blk = build_csp(block_policy="fixed", block_size=32, block_target=0.5, device="cpu", block_graph=graph)
    with torch.no_grad():
        want = dense.backbone(x)
        xw = blockcopy.to_tensorwrapper(x)
        xw.process_temporal_features(None)
        grid = torch.ones(1, 1, 4, 8, dtype=torch.bool)
        got = blk.backbone(xw.to_blocks(grid, grid))
        for w, g in zip(want, got):
            full = g.combine().to_tensor()
            assert float((full - w).abs().max()) <= 1e-4 * max(1.0, float(w.abs().max()))

        neck_maps = []
        hook = blk.neck.register_forward_hook(lambda mod, inp, out: neck_maps.append(out[0].combine().to_tensor().clone()))
        blk.reset_temporal()
        for t in range(4):
            blk.simple_test(x)
        hook.remove()
        assert len(neck_maps) == 4 and neck_maps[0].shape == (1, 768, 32, 64)
        for t in range(1, 4):
            assert float((neck_maps[t] - neck_maps[0]).abs().max()) <= 1e-4 * max(1.0, float(neck_maps[0].abs().max())), t
        assert blk.policy.stats.exec == 32 + 3 * 16

        allb = build_csp(block_policy="all", block_size=32, device="cpu", block_graph=graph)
        allb.reset_temporal()
        allb.simple_test(x)
        first = [m.clone() for m in allb.head_out]
        allb.simple_test(x)
        for a, b_ in zip(first, allb.head_out):
            assert a.shape[-2:] == (32, 64) and float((a - b_).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))


def test_detections_and_reference_output_format(oracle_backend):
    from bc_workloads.csp import build_csp

    blk = build_csp(block_policy="fixed", block_size=32, block_target=0.5, device="cpu", results="numpy")
    x = _tiny_inputs(1)
    blk.reset_temporal()
    out = blk.simple_test(x)
    assert isinstance(out, list) and len(out) == 1 and isinstance(out[0], np.ndarray) and out[0].shape[1] == 5
    assert out[0].shape[0] <= 100 and (out[0][:, 4] > 0.1).all()
    assert (out[0][:, 2] >= out[0][:, 0]).all() and (out[0][:, 0] >= 0).all() and (out[0][:, 2] <= 255).all()
    # num_exec == 0 returns the cached result object
    blk.policy.block_target = 0.0
    prev = blk.policy_meta["outputs"]
    blk.simple_test(x)
    assert blk.policy_meta["num_exec"] == 0 and blk.policy_meta["outputs"] is prev


def test_three_head_branches_share_one_halo_gather(oracle_backend):
    """The 768-channel head input is padded once for the three 3x3 branches."""
    import blockcopy.backend as bk
    from bc_workloads.csp import build_csp

    be = bk.get_backend()
    calls = []
    orig = be.pad_ring

    def counting(data, *a, **k):
        calls.append(tuple(data.shape))
        return orig(data, *a, **k)

    be.pad_ring = counting
    try:
        blk = build_csp(block_policy="all", block_size=32, device="cpu")
        blk.reset_temporal()
        blk.simple_test(_tiny_inputs(2))
    finally:
        be.pad_ring = orig
    assert sum(1 for s in calls if s[1] == 768) == 1, calls


def test_rl_objectdetection_policy_drives_the_detector(oracle_backend):
    """The reference's detector config uses the `rl_objectdetection` policy (csp_r50_clip_blockcopy_030.py:5-17): its
    output representation and information gain are built from the per-class numpy box lists the detector returns."""
    import random

    from bc_workloads.csp import build_csp

    torch.manual_seed(0)
    random.seed(0)
    blk = build_csp(block_policy="rl_objectdetection", block_size=32, block_target=0.3, device="cpu", results="numpy",
                    block_train_interval=2)
    assert blk.policy.net.training
    w0 = blk.policy.net.backbone.conv1.weight.detach().clone()
    blk.reset_temporal()
    execs = []
    for t in range(4):
        out = blk.simple_test(_tiny_inputs(t))
        assert isinstance(out, list) and out[0].shape[1] == 5
        execs.append(blk.policy_meta["num_exec"])
        assert blk.policy_meta["output_repr"].shape == (1, 1, 128, 256)
    assert execs[0] == 32 and all(e % 2 == 0 for e in execs)
    assert "information_gain" in blk.policy_meta and blk.policy_meta["information_gain"].shape == (1, 1, 128, 256)
    assert not torch.equal(w0, blk.policy.net.backbone.conv1.weight)      # the policy trained online


# ------------------------------------------------------------------------------------------------------------------
# Reference-generated fixtures for the detector op classes: group_norm batched trick, per-tile
# conv_transpose2d, dilation-2 halo, blockcopy.to_tensor inside the head.  tests/golden/{tinycsp,csp_r50}.npz were
# produced by the REFERENCE TensorWrapper (oracle/gen_golden.py gen_tinycsp / gen_csp_r50) for the same torch modules.
@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("engine", ["fused", "reference"])
def test_tinycsp_matches_reference(golden_dir, oracle_backend, engine, graph):
    import tinycsp
    from blockcopy.core import tensorwrapper as tw
    from common import load_golden, run_golden_detector_clip

    if engine == "reference" and graph:
        pytest.skip("graph replay exists for the fused engine only")
    G, cfg = load_golden(golden_dir, "tinycsp.npz")
    try:
        errs, fs = run_golden_detector_clip(G, cfg, tinycsp.seed_weights(tinycsp.tinycsp_arch()), "cpu", engine, graph)
    finally:
        tw.set_engine("fused")
    assert max(errs) <= 2e-5, errs
    assert fs and all(e == 0.0 for e in fs)


@pytest.mark.parametrize("engine,cl", [("fused", False), ("fused", True), ("reference", False)])
def test_csp_r50_matches_reference(golden_dir, oracle_backend, engine, cl):
    """This repo's CSP-ResNet50 restatement (bc_workloads/csp.py) through this repo's engine vs the SAME modules through
    the reference's TensorWrapper (128x256, block 32, masks all / half / one / all-but-one)."""
    import json
    from bc_workloads import seeded
    from bc_workloads.csp import CSP
    from blockcopy.core import tensorwrapper as tw
    from common import load_golden, run_golden_detector_clip

    G, cfg = load_golden(golden_dir, "csp_r50.npz")
    det = CSP()
    det.load_state_dict(seeded.name_seeded_state_dict(dict(det.state_dict())), strict=True)
    try:
        errs, fs = run_golden_detector_clip(G, cfg, (det.backbone, det.neck, det.bbox_head), "cpu", engine, 0, channels_last=cl)
    finally:
        tw.set_engine("fused")
    assert max(errs) <= 5e-5, errs
    assert fs and all(e == 0.0 for e in fs)
    calls = json.loads(bytes(G["kernel_calls"]).decode())
    assert calls["pad"] == 21 * cfg["n_frames"]     # 16 bottleneck 3x3 + stem conv + stem pool + 3 head convs, per frame


def test_vectorised_detection_rewards_equal_the_loop_restatement():
    """build_instance_mask / build_instance_mask_iou_gain without per-box loops == the box-by-box
    restatement of reference policy/information_gain.py:55-108, bit for bit, on random and adversarial detections
    (ties, collapsed boxes at half resolution, no matches, empty frames, several classes)."""
    import policy_oracle as L
    from blockcopy.policy import information_gain as V

    rng = np.random.default_rng(3)

    def dets(n, H=96, W=160):
        x1, y1 = rng.integers(0, W - 8, n), rng.integers(0, H - 8, n)
        w, h = rng.integers(1, 40, n), rng.integers(1, 60, n)
        b = np.stack([x1, y1, np.minimum(x1 + w, W - 1), np.minimum(y1 + h, H - 1), rng.random(n) * 0.9 + 0.1], 1).astype(np.float32)
        return b

    cases = [(dets(30), dets(25)), (dets(1), dets(0)), (dets(0), dets(7)), (dets(0), dets(0)), (dets(120), dets(150))]
    a = dets(12)
    shifted = a.copy(); shifted[:, [0, 2]] += 2                      # heavy overlaps
    cases += [(a, a.copy()), (a, shifted), (np.repeat(a[:3], 3, 0), a[:3])]   # exact matches (gain 0), duplicates / ties
    tiny = a.copy(); tiny[:, 2] = tiny[:, 0] + 1; tiny[:, 3] = tiny[:, 1] + 1   # collapse to empty at SUBSAMPLE 2
    cases.append((tiny, a))
    size = (1, 1, 96, 160)
    for cur, prev in cases:
        want = L.build_instance_mask_iou_gain([[cur]], [[prev]], size)
        got = V.build_instance_mask_iou_gain([[cur]], [[prev]], size)
        assert got.shape == want.shape == size and torch.equal(got, want)
        assert torch.equal(V.build_instance_mask([[cur]], size), L.build_instance_mask([[cur]], size))
    # two classes: both paint channel 0 of the gain map (reference quirk); the score mask keeps its per-channel rule
    size2 = (1, 2, 96, 160)
    c2, p2 = [[dets(9), dets(11)]], [[dets(10), dets(4)]]
    assert torch.equal(V.build_instance_mask_iou_gain(c2, p2, size2), L.build_instance_mask_iou_gain(c2, p2, size2))
    assert torch.equal(V.build_instance_mask(c2, size2), L.build_instance_mask(c2, size2))


def test_detection_rewards_match_the_reference(golden_dir):
    """tests/golden/detgain.npz: InformationGainObjectDetection of the REFERENCE (policy/information_gain.py:43-108) run in the build
    container on seeded detections -- random sets, empty frames, exact matches (gain 0), heavy overlaps, duplicates / ties.  Both the
    box-by-box restatement (tests/policy_oracle.py) and the product's vectorised form must reproduce its maps bit for bit."""
    import policy_oracle as L
    from blockcopy.policy import information_gain as V

    G = np.load(os.path.join(golden_dir, "detgain.npz"))
    cfg = json.loads(bytes(G["cfg"]).decode())
    size = (1, cfg["num_classes"], cfg["H"], cfg["W"])
    ig = V.InformationGainObjectDetection(num_classes=cfg["num_classes"])
    frame = torch.zeros(1, 3, cfg["H"], cfg["W"])
    for k in range(cfg["n_cases"]):
        cur, prev = G[f"cur{k}"], G[f"prev{k}"]
        want_gain, want_repr = torch.from_numpy(G[f"gain{k}"]), torch.from_numpy(G[f"repr{k}"])
        assert torch.equal(L.build_instance_mask_iou_gain([[cur]], [[prev]], size), want_gain), k
        assert torch.equal(L.build_instance_mask([[cur]], size), want_repr), k
        pm = {"inputs": frame, "outputs": [[cur]], "outputs_prev": [[prev]]}
        assert torch.equal(ig(pm), want_gain), k
        assert torch.equal(ig.get_output_repr(pm), want_repr), k


# ------------------------------------------------------------------------------------------------------------------
# The model, its decode and NMS against the REFERENCE'S OWN mmdet modules (tests/golden/csp_ref_modules.npz, generated by
# oracle/gen_golden.py gen_csp_ref_modules from detectors/csp_blockcopy.py, backbones/resnet.py, necks/csp_neck.py,
# anchor_heads/csp_head.py, core/bbox/transforms.py, core/post_processing/bbox_nms.py loaded by path).
def _csp_ref(golden_dir):
    return np.load(os.path.join(golden_dir, "csp_ref_modules.npz"))


def test_state_dict_keys_equal_the_reference_detector(golden_dir):
    """Key set AND shapes of the reference's CSPBlockCopy (C5 config, policy parameters aside) == this repo's CSP: a Pedestron
    checkpoint loads with strict=True."""
    from bc_workloads.csp import CSP

    want = json.loads(bytes(_csp_ref(golden_dir)["state_dict_keys"]).decode())
    have = {k: list(v.shape) for k, v in CSP().state_dict().items()}
    assert len(want) == 344 and have == want, sorted(set(want) ^ set(have))[:10]
    assert list(have) == list(want), "registration order differs (matters for BN folding and name-seeded fills)"


@pytest.mark.parametrize("tag,engine,graph,fold", [("a", "fused", 0, False), ("a", "fused", 1, True), ("a", "reference", 0, False), ("b", "fused", 0, True)])
def test_csp_matches_reference_modules(oracle_backend, golden_dir, tag, engine, graph, fold):
    """Own CSP-ResNet50 + manager + decode on the checker backend vs the reference detector's head maps (<= 5e-5), packed neck output,
    and boxes: decoding the fixture's maps reproduces the reference's post-NMS boxes (count, order, <= 1e-4 px / score)."""
    from common import run_csp_ref_clip

    w = run_csp_ref_clip(_csp_ref(golden_dir), tag, "cpu", engine, graph, fold_bn=fold, repeats=2 if graph else 1)
    assert w["maps"] <= 5e-5 and w["neck"] <= 5e-5, w
    assert 0 <= w["decode"] <= 1e-4, w
    assert w["e2e_match"] >= 0.9, w


def test_dense_csp_matches_the_reference_dense_detector(oracle_backend, golden_dir):
    """detectors/csp.py + single_stage.py:61-70 (no block path) on frame 0 of clip a: maps and boxes of this repo's dense CSP."""
    from bc_workloads import seeded
    from bc_workloads.csp import CSP

    G = _csp_ref(golden_dir)
    cfg = json.loads(bytes(G["a_cfg"]).decode())
    det = CSP()
    det.load_state_dict(seeded.csp_reference_weights(dict(det.state_dict())), strict=True)
    det.eval()
    x = seeded.synthetic_frame(cfg["frame_seed0"], (cfg["N"], 3, cfg["H"], cfg["W"]))
    with torch.no_grad():
        maps = det.head_maps(x)
        dets, _ = det.bbox_head.get_bboxes(*[torch.from_numpy(G[f"dense_{k}"]) for k in ("cls", "reg", "offset")], img_shape=(cfg["H"], cfg["W"]))
    for k, m in zip(("cls", "reg", "offset"), maps):
        want = torch.from_numpy(G[f"dense_{k}"])
        assert float((m - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), k
    want = torch.from_numpy(G["dense_boxes"])
    assert dets.shape == want.shape and float((dets - want).abs().max()) <= 1e-4
