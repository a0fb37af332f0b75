"""Components either side of the hot path: Cityscapes clip reader, StreamSegMetrics, GMACs
counter -- against tests/golden/io_metrics.npz, produced by the REFERENCE's CityscapesVid / StreamSegMetrics /
flopscounter (oracle/gen_golden.py gen_io_metrics)."""
import json
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "io_metrics.npz"))


def test_clip_reader_matches_reference(G, tmp_path):
    from bc_workloads import cityscapes as cs

    cfg = json.loads(bytes(G["io_cfg"]).decode())
    assert np.allclose(cs.MEAN, cfg["mean"]) and np.allclose(cs.STD, cfg["std"]) and cs.FINE_CLASSES == cfg["fine_classes"]
    root = str(tmp_path)
    rels = cs.write_synthetic_tree(root, cfg["split"], cfg["cities"], cfg["clips_per_city"], cfg["clip_length"], tuple(cfg["size"]), cfg["seed"])
    assert sorted(rels) == cfg["relpaths"]
    plain = lambda img, lbl: (np.array(img, dtype=np.uint8), lbl)     # noqa: E731
    ds = cs.CityscapesClips(root, split=cfg["split"], transform=plain, clip_length=cfg["clip_length"], has_labels=True)
    assert len(ds) == len(rels)
    by_rel = {ds.relative_dirs[i]: i for i in range(len(ds))}
    for k, rel in enumerate(cfg["relpaths"]):
        frames, target, meta = ds[by_rel[rel]]
        assert meta == {"relpath": rel}
        assert np.array_equal(np.stack(frames), G[f"clip{k}_frames"])       # chronological order, labelled frame last
        assert np.array_equal(np.asarray(target), G[f"clip{k}_target"])     # ids -> train ids
    # chronological order: the labelled frame comes from leftImg8bit, the others count DOWN from it in leftImg8bit_sequence
    paths = ds.clip_paths(by_rel[cfg["relpaths"][0]])
    assert paths[-1].endswith("leftImg8bit/val/aachen/aachen_000000_000019_leftImg8bit.png")
    assert [os.path.basename(p) for p in paths[:-1]] == [f"aachen_000000_{n:06d}_leftImg8bit.png" for n in (16, 17, 18)]
    assert all("leftImg8bit_sequence" in p for p in paths[:-1])
    # unlabelled mode and the default normalising transform
    ds2 = cs.CityscapesClips(root, split=cfg["split"], transform=cs.normalize_transform((16, 32)), clip_length=2, has_labels=False)
    frames, target, _ = ds2[0]
    assert target == 0 and len(frames) == 2 and frames[0].shape == (3, 16, 32) and frames[0].dtype == torch.float32
    raw = np.asarray(__import__("PIL.Image", fromlist=["Image"]).open(ds2.clip_paths(0)[-1]).convert("RGB").resize((32, 16), 2), dtype=np.float32) / 255
    want = (torch.from_numpy(raw).permute(2, 0, 1) - torch.tensor(cs.MEAN).view(3, 1, 1)) / torch.tensor(cs.STD).view(3, 1, 1)
    assert torch.allclose(frames[-1], want, atol=1e-6)
    with pytest.raises(RuntimeError, match="Dataset not found"):
        cs.CityscapesClips(os.path.join(root, "nope"), split="val")
    with pytest.raises(ValueError):
        cs.CityscapesClips(root, split="dev")


def test_label_coding_matches_reference(G):
    from bc_workloads import cityscapes as cs

    assert np.array_equal(cs.encode_target(np.arange(-1, 34)), G["encode_all_ids"])
    assert np.array_equal(cs.decode_target(np.array(list(range(19)) + [255])), G["decode_all_train_ids"])
    assert np.array_equal(cs.encode_target_test(np.arange(19)), G["encode_test_all"])


def test_stream_seg_metrics_match_reference(G):
    from bc_workloads.metrics import cityscapes_metrics

    want = json.loads(bytes(G["m_results"]).decode())
    m = cityscapes_metrics()
    for u in range(3):
        m.update(G[f"m_lt{u}"], G[f"m_lp{u}"])
        got = m.get_results()
        for k, v in want[u].items():
            if k == "Class IoU":
                assert list(got[k]) == list(v)
                for name in v:
                    assert (np.isnan(v[name]) and np.isnan(got[k][name])) or got[k][name] == pytest.approx(v[name], rel=1e-12), (u, name)
            else:
                assert got[k] == pytest.approx(v, rel=1e-12), (u, k)
    assert np.array_equal(m.confusion_matrix, G["m_confusion"])
    for a, key in ((m.ious, "m_ious"), (m.ious_sum, "m_ious_sum"), (m.accs, "m_accs"), (m.accs_sum, "m_accs_sum")):
        assert np.allclose(np.array(a), G[key], rtol=1e-12, equal_nan=True)
    assert "Mean IoU" in m.to_str(got) and "Class IoU" not in m.to_str(got)
    m.reset()
    assert m.confusion_matrix.sum() == 0


def test_gmacs_counter_matches_reference(G, oracle_backend):
    """Same numbers as the reference's flopscounter on the reference's SwiftNet-RN18: dense, and under block execution
    where a conv is charged for the EXECUTED tiles only (output.shape[0] = n_exec, flopscounter.py:341-373)."""
    from bc_workloads import harness, seeded
    from bc_workloads.gmacs import GMACsCounter
    from common import make_forced_policy, scenario_grids as _scenario_grids

    g = json.loads(bytes(G["gmacs"]).decode())
    shape = (g["N"], 3, g["H"], g["W"])
    frames = [seeded.synthetic_frame(g["frame_seed0"] + t, shape) for t in range(4)]
    dense = harness.build_model("resnet18", block_policy="static", device="cpu", fold_bn=False)
    with torch.no_grad(), GMACsCounter(dense) as c:
        assert c.compute_average_flops_cost() == 0
        dense(frames[0])
        dense(frames[1])
    assert list(c.compute_average_flops_cost()) == [g["dense_avg"], g["dense_frames"]]

    grids = _scenario_grids(g["N"], g["H"] // g["block_size"], g["W"] // g["block_size"], g["grid_seed"])[:4]
    assert [int(x.sum()) for x in grids] == g["exec"]
    blk = harness.build_model("resnet18", block_policy="all", block_size=g["block_size"], device="cpu")
    blk.policy = make_forced_policy(g["block_size"], grids)
    blk.reset_temporal()
    with torch.no_grad(), GMACsCounter(blk) as c:
        for t in range(4):
            blk(frames[t])
            assert c.compute_total_flops_cost()[0] == g["block_total_after_frame"][t], t
    assert list(c.compute_average_flops_cost()) == [g["block_avg"], g["block_frames"]]
    assert g["block_avg"] < g["dense_avg"]
