"""CPU tests of the host side of the engine (state machine, op routing, index tables, ring-cache bookkeeping, policies)
against the golden fixtures produced by the REFERENCE Python.  The block ops are served by the injected oracle
backend (tests/oracle_backend.py); the product itself has no CPU path (see test_no_cpu_fallback)."""
import json
import os

import numpy as np
import pytest
import torch

from common import golden_logit_error, load_golden, make_forced_policy, run_golden_clip


class TinyNet(torch.nn.Module):
    """Same tiny conv/pool net as oracle/gen_golden.py::TinyNet (pads 1 and 2, tiles 8 -> 4 -> 2)."""

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


def tiny_grids():
    from bc_workloads import seeded

    N, GH, GW = 2, 2, 3
    gs = [torch.ones(N, 1, GH, GW, dtype=torch.bool)]
    for s, n in ((11, 6), (12, 3), (13, 1), (14, 11)):
        gs.append(seeded.fixed_fraction_grid(s, N, GH, GW, n))
    return gs


def test_index_tables_match_reference(golden_dir, oracle_backend):
    """BlockFeatures._process_grid vs the reference's get_grid_mappings / transfer_idx (core/tensorwrapper.py:108-178)."""
    from blockcopy.core.tensorwrapper import BlockFeatures

    G = np.load(os.path.join(golden_dir, "index_tables.npz"))
    for c in json.loads(bytes(G["meta"]).decode()):
        prev = None
        for f in range(c["frames"]):
            k = f"c{c['case']}_f{f}"
            grid = torch.from_numpy(G[k + "_grid"])
            bf = BlockFeatures("cpu", engine="reference")
            bf._process_grid(grid, prev, grid)
            assert np.array_equal(bf._grid_idx.numpy(), G[k + "_grid_idx"])
            assert np.array_equal(bf._mapping_exec.numpy(), G[k + "_mapping_exec"])
            if prev is not None:
                assert np.array_equal(bf._transfer_idx.numpy(), G[k + "_transfer_idx"])
            prev = bf


def test_product_grid_tables_host_matches_reference(golden_dir):
    """The library's host-side table builder (bc_grid_tables_host) against the same fixtures -- no GPU needed."""
    import blockcopy.backend as bk

    lib_backend = bk.HipBackend()
    G = np.load(os.path.join(golden_dir, "index_tables.npz"))
    for c in json.loads(bytes(G["meta"]).decode()):
        prev = None
        for f in range(c["frames"]):
            k = f"c{c['case']}_f{f}"
            g8 = G[k + "_grid"].astype(np.uint8).reshape(-1)
            n = g8.size
            gi, mp, tr = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
            n_exec = lib_backend.grid_tables_host(g8, gi, mp, prev, tr if prev is not None else None)
            assert np.array_equal(gi.reshape(G[k + "_grid_idx"].shape), G[k + "_grid_idx"])
            assert np.array_equal(mp[:n_exec], G[k + "_mapping_exec"])
            if prev is not None:
                assert np.array_equal(tr[:n - n_exec], G[k + "_transfer_idx"])
            prev = gi


@pytest.mark.parametrize("engine", ["fused", "reference"])
def test_tinynet_matches_reference(golden_dir, oracle_backend, engine):
    import blockcopy
    from blockcopy.core import tensorwrapper as tw
    from blockcopy.core.argparser import default_settings
    from bc_workloads import seeded

    G = np.load(os.path.join(golden_dir, "ops_tinynet.npz"))
    tw.set_engine(engine)
    try:
        net = TinyNet()
        net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()))
        model = blockcopy.BlockCopyModel(net.eval(), default_settings(block_policy="all", block_size=8))
        grids = tiny_grids()
        model.policy = make_forced_policy(8, grids)
        model.reset_temporal()
        with torch.no_grad():
            for t in range(len(grids)):
                y = model(seeded.synthetic_frame(100 + t, (2, 3, 16, 24)))
                assert torch.equal(y, torch.from_numpy(G[f"net_out{t}"])), (engine, t)
    finally:
        tw.set_engine("fused")


@pytest.mark.parametrize("engine", ["fused", "reference"])
@pytest.mark.parametrize("name", ["swiftnet_rn18_a.npz", "swiftnet_rn18_n2.npz", "swiftnet_rn50_a.npz", "swiftnet_rn18_c.npz"])
def test_swiftnet_matches_reference_on_cpu(golden_dir, oracle_backend, name, engine):
    """Whole SwiftNet clip through OUR BlockCopyModel/TensorWrapper/SwiftNet/BN-fold vs the reference's logits.
    Includes an all-skipped frame (num_exec == 0 returns the cached output) and single-tile / all-but-one masks."""
    from blockcopy.core import tensorwrapper as tw

    G, cfg = load_golden(golden_dir, name)
    try:
        errs, fs_errs = run_golden_clip(G, cfg, "cpu", engine)
    finally:
        tw.set_engine("fused")
    assert max(errs) <= 2e-5, errs          # same oneDNN convs; tiny differences from bilinear vs trilinear and fold order
    assert all(e == 0.0 for e in fs_errs)   # frame_state: pure copies


def test_state_dict_keys_match_reference(golden_dir):
    """Our SwiftNet restatement exposes exactly the reference's parameter names and shapes."""
    from bc_workloads.swiftnet import build_swiftnet

    with open(os.path.join(golden_dir, "swiftnet_keys.json")) as f:
        want = json.load(f)
    for backbone in ("resnet18", "resnet50"):
        sd = build_swiftnet(backbone).state_dict()
        got = {k: list(v.shape) for k, v in sd.items()}
        assert got == want[backbone]
        assert list(got) == list(want[backbone])   # same order too


def test_num_exec_zero_returns_cached_output_object(oracle_backend):
    import blockcopy
    from blockcopy.core.argparser import default_settings

    net = TinyNet().eval()
    model = blockcopy.BlockCopyModel(net, default_settings(block_policy="none", block_size=8))
    x = torch.randn(1, 3, 16, 16)
    with torch.no_grad():
        y0 = model(x)
        y1 = model(x)   # PolicyNone still executes frame 2 (outputs_prev is None until then, reference policy.py:188)
        y2 = model(x)
    assert model.policy_meta["num_exec"] == 0
    assert y2 is y1 and y1 is not y0


def test_first_frame_must_execute_everything(oracle_backend):
    import blockcopy

    x = blockcopy.to_tensorwrapper(torch.randn(1, 3, 16, 16))
    x.process_temporal_features(None)
    grid = torch.zeros(1, 1, 2, 2, dtype=torch.bool)
    grid[0, 0, 0, 0] = True
    with pytest.raises(AssertionError, match="first run should execute all blocks"):
        x.to_blocks(grid, grid)


def test_routing_errors_and_flags(oracle_backend):
    import blockcopy

    x = blockcopy.to_tensorwrapper(torch.randn(1, 4, 16, 16))
    x.process_temporal_features(None)
    grid = torch.ones(1, 1, 2, 2, dtype=torch.bool)
    b = x.to_blocks(grid, grid)
    assert blockcopy.is_block(b) and blockcopy.is_tensorwrapper(b) and b.block_size == 8 and b.shape == (4, 4, 8, 8)
    assert not blockcopy.is_block(x) and x.block_size == -1
    with pytest.raises(AttributeError, match="not supported"):
        torch.nn.functional.adaptive_avg_pool2d(b, 1)
    with pytest.raises(AttributeError, match="not supported"):
        b.view(-1)
    with pytest.raises(NotImplementedError, match="equal paddings"):
        torch.nn.functional.conv2d(b, torch.randn(4, 4, 3, 3), None, 1, (1, 2))
    with pytest.raises(AttributeError, match="already split"):
        b._split(8)
    with pytest.raises(AttributeError, match="Not split"):
        x.combine()
    y = torch.relu(b) + 1
    assert blockcopy.is_block(y) and y.get_grid_idx() is b.get_grid_idx()
    d = blockcopy.to_tensor({"a": [y], "b": (y, 3)})
    assert type(d["a"][0]) is torch.Tensor and d["a"][0].shape == (1, 4, 16, 16) and d["b"][1] == 3
    # group_norm statistics run over all executed tiles of the frame
    gn = torch.nn.functional.group_norm(b, 2)
    dense_gn = torch.nn.functional.group_norm(b.combine().to_tensor(), 2)
    assert torch.allclose(blockcopy.to_tensor(gn), dense_gn, atol=1e-5)


def test_op_order_divergence_is_detected(oracle_backend):
    import blockcopy

    def frame(channels):
        x = blockcopy.to_tensorwrapper(torch.randn(1, channels, 16, 16))
        return x

    x = frame(4)
    feats = x.process_temporal_features(None)
    grid = torch.ones(1, 1, 2, 2, dtype=torch.bool)
    torch.nn.functional.conv2d(x.to_blocks(grid, grid), torch.randn(4, 4, 3, 3), padding=1)
    x2 = frame(5)
    x2.process_temporal_features(feats)
    with pytest.raises(AssertionError, match="same op sequence"):
        torch.nn.functional.conv2d(x2.to_blocks(grid, grid), torch.randn(5, 5, 3, 3), padding=1)


def test_policies(oracle_backend):
    import random

    from blockcopy.core.argparser import default_settings
    from blockcopy.policy.policy import build_policy_from_settings

    x = torch.randn(1, 3, 64, 128)
    meta = {"inputs": x, "outputs": None, "outputs_prev": None}
    pol = build_policy_from_settings(default_settings(block_policy="fixed", block_size=16, block_target=0.5))
    m = pol(dict(meta))
    assert m["num_exec"] == m["num_total"] == 32 and m["grid"].dtype == torch.bool and m["grid"].shape == (1, 1, 4, 8)
    m2 = pol(dict(meta, outputs=torch.zeros(1)))
    assert m2["num_exec"] == 16 and torch.equal(m2["grid"], m2["grid_host"])
    random.seed(0)
    torch.manual_seed(0)
    pol = build_policy_from_settings(default_settings(block_policy="random", block_size=16))
    m3 = pol(dict(meta, outputs=torch.zeros(1), outputs_prev=torch.zeros(1)))
    assert m3["num_exec"] % 2 == 0 and 0 < m3["num_exec"] <= 32   # quantised to total/16 = 2
    assert abs(pol.stats.get_exec_percentage() - m3["perc_exec"]) < 1e-9
    with pytest.raises(AssertionError, match="not a multiple of block size"):
        pol(dict(meta, inputs=torch.randn(1, 3, 60, 128)))


def test_dense_map_routes_small_cout_convs_through_the_backend(oracle_backend):
    """to_tensor's DenseMap (core/tensorwrapper.py): a 3x3 / padding 1 conv to <= 4 channels goes to the backend's pred3x3 (here the
    checker's library conv, spied), anything else -- wider convs, strided ones, parameters under autograd, elementwise ops -- takes the
    stock path, and every result is a plain tensor."""
    import torch.nn.functional as F
    import blockcopy.backend as bk
    from blockcopy.core import tensorwrapper as tw

    be = bk.get_backend()
    calls = []
    orig = be.pred3x3
    be.pred3x3 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    g = torch.Generator().manual_seed(3)
    x = torch.randn((1, 64, 12, 20), generator=g).contiguous(memory_format=torch.channels_last)
    dm = x.as_subclass(tw.DenseMap)
    w, b = torch.randn((2, 64, 3, 3), generator=g) * 0.05, torch.randn(2, generator=g)
    y = F.conv2d(dm, w, b, padding=1)
    assert calls == [1] and type(y) is torch.Tensor and torch.allclose(y, F.conv2d(x, w, b, padding=1), atol=1e-5)
    conv = torch.nn.Conv2d(64, 1, 3, padding=1)
    assert type(conv(dm)) is torch.Tensor and calls == [1]                     # parameters that want gradients: stock conv
    with torch.no_grad():
        y2 = conv(dm)
    assert calls == [1, 1] and type(y2) is torch.Tensor and torch.allclose(y2, conv(x).detach(), atol=1e-5)
    for other in (F.conv2d(dm, torch.randn((8, 64, 3, 3), generator=g), padding=1), F.conv2d(dm, w, b, padding=1, stride=2),
                  F.conv2d(dm, torch.randn((2, 64, 1, 1), generator=g)), dm * 2 + 1, torch.relu(dm), dm[0], dm.mean()):
        assert type(other) is torch.Tensor
    assert calls == [1, 1] and torch.equal(dm * 2 + 1, x * 2 + 1)


def test_no_cpu_fallback():
    """Without an injected checker the package's only backend is the HIP library, which refuses CPU tensors."""
    import blockcopy.backend as bk

    prev = bk.set_backend(None)
    try:
        be = bk.get_backend()
        assert be.name == "hip"
        with pytest.raises(AssertionError, match="GPU"):
            be.split(torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 8, 8), torch.zeros(1, dtype=torch.int32), torch.zeros(1, 1, 2, 2, dtype=torch.int32))
        with pytest.raises(bk.BlockCopyBackendError, match="no fallback"):
            bk.load_library("/nonexistent/libblockcopy_hip.so")
    finally:
        bk.set_backend(prev)


@pytest.mark.parametrize("name", ["swiftnet_rn18_a.npz", "swiftnet_rn18_n2.npz"])
def test_persistent_state_pipeline_matches_reference_on_cpu(golden_dir, oracle_backend, name):
    """The graph-capturable body (persistent ring caches / dense maps, static index tables) run eagerly on CPU:
    same logits as the reference over two consecutive clips that share the persistent buffers."""
    G, cfg = load_golden(golden_dir, name)
    errs, fs_errs = run_golden_clip(G, cfg, "cpu", "fused", graph=1, repeats=2)
    assert len(errs) == 2 * cfg["n_frames"] and max(errs) <= 2e-5, errs
    assert all(e == 0.0 for e in fs_errs)


def _rl_model(cfg, device, graph=0):
    import blockcopy
    from blockcopy.core.argparser import default_settings
    from bc_workloads import seeded
    from bc_workloads.bn_fold import fold_batchnorm
    from bc_workloads.swiftnet import build_swiftnet

    net = build_swiftnet("resnet18")
    net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()), strict=True)
    net.eval()
    st = default_settings(block_policy="rl_semseg", block_size=cfg["block_size"], block_target=cfg["block_target"],
                          block_train_interval=cfg["train_interval"], block_graph=graph)
    model = blockcopy.BlockCopyModel(net, st)
    model.policy.net.load_state_dict(seeded.name_seeded_state_dict(model.policy.net.state_dict()))
    model = fold_batchnorm(model.to(device))
    assert model.policy.net.training   # the policy net trains online; only the base model is in eval mode
    return model


@pytest.mark.parametrize("fixture", ["rl_semseg_run.npz", "rl_semseg_c3.npz"])
def test_rl_policy_loop_matches_reference(golden_dir, oracle_backend, fixture):
    """Config C3 plumbing: PolicyTrainRL + PolicyNet + InformationGainSemSeg + RMSprop with the reference's seeds reproduce its
    sampled grids, probabilities, information gain, running cost and weight update (rl_semseg_run.npz: 4 frames at 128x256;
    rl_semseg_c3.npz: C3 at its FULL size -- 1024x2048, block 128, target 0.3, train_interval 3, 5 frames).

    RMSprop's first steps are sign-like (step ~ lr / sqrt(1 - alpha) whatever the gradient magnitude), so weights whose
    gradient is ~0 may step the other way under 1e-6 input noise: after the first update the comparison is on the
    update direction (cosine / sign agreement) and on loose probabilities; before it, everything is tight."""
    import random

    from bc_workloads import seeded

    G, cfg = load_golden(golden_dir, fixture)
    model = _rl_model(cfg, "cpu")
    w0 = model.policy.net.state_dict()["backbone.conv1.weight"].clone()
    torch.manual_seed(0)
    random.seed(0)
    model.reset_temporal()
    # Frames decided BEFORE the first optimiser step must reproduce the reference's samples exactly (same RNG stream, same
    # probabilities).  After a step the probabilities agree only loosely (see above), so a borderline tile may be sampled the other
    # way: from then on the full-size clip is teacher-forced with the fixture's grid (the small clip happens to stay exact).
    frame = {"t": 0}
    quantize = model.policy.quantize_number_exec_grid
    forced_from = cfg["train_interval"] if fixture == "rl_semseg_c3.npz" else cfg["n_frames"]
    model.policy.quantize_number_exec_grid = lambda g: torch.from_numpy(G[f"grid{frame['t']}"]).clone() if frame["t"] >= forced_from else quantize(g)
    with torch.no_grad():
        for t in range(cfg["n_frames"]):
            frame["t"] = t
            y = model(seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"])))
            pm = model.policy_meta
            assert np.array_equal(pm["grid"].numpy(), G[f"grid{t}"]), t          # same Bernoulli samples + quantisation
            assert golden_logit_error(G, cfg, t, y) <= 2e-5 * max(1.0, float(G[f"logits{t}_absmax"]) if f"logits{t}_absmax" in G.files else 1.0)
            if t > 0:
                tight = t < cfg["train_interval"]   # no optimiser step has happened yet
                assert np.allclose(pm["grid_probs"].detach().numpy(), G[f"grid_probs{t}"], rtol=2e-3 if tight else 0, atol=1e-6 if tight else 0.05)
            if f"information_gain{t}" in G.files:
                assert np.allclose(pm["information_gain"].detach().numpy(), G[f"information_gain{t}"], rtol=1e-3, atol=1e-6)
            assert abs(model.policy.running_cost - float(G[f"running_cost{t}"])) < 1e-12
    w1 = model.policy.net.state_dict()["backbone.conv1.weight"]
    d_me, d_ref = (w1 - w0).numpy().ravel(), (G["policy_conv1_after"] - w0.numpy()).ravel()
    assert np.linalg.norm(d_ref) > 0.05   # the policy did train (two RMSprop steps)
    assert float(d_me @ d_ref / np.linalg.norm(d_me) / np.linalg.norm(d_ref)) > 0.95
    assert float(np.mean(np.sign(d_me) == np.sign(d_ref))) > 0.98
    assert set(pm) >= {"frame_state", "grid", "grid_log_probs", "grid_probs", "information_gain", "inputs", "num_exec", "num_total",
                       "output_repr", "outputs", "outputs_prev", "perc_exec"}


@pytest.mark.parametrize("name", ["swiftnet_rn18_a.npz", "swiftnet_rn18_n2.npz"])
def test_channels_last_model_matches_reference_on_cpu(golden_dir, oracle_backend, name):
    """A channels-last SwiftNet drives the engine with channels-last packed tiles (everything after the first conv):
    the layout plumbing (packed/dense/persistent buffers keep the producer's memory format) must not change logits."""
    import blockcopy
    from blockcopy.backend import is_nhwc
    from blockcopy.core.argparser import default_settings
    from bc_workloads import seeded
    from bc_workloads.bn_fold import fold_batchnorm
    from bc_workloads.swiftnet import build_swiftnet

    import blockcopy.backend as bk

    G, cfg = load_golden(golden_dir, name)
    # the engine's MI355X-specific routes must be taken on this model: residual gather (by-product), fused halo+pool
    chk, hits = bk.get_backend(), {"pad_ring_add": 0, "maxpool3x3s2_ring": 0, "conv3x3_ring": 0, "head1x1_scatter": 0, "combine_copy": 0}
    for meth in hits:
        orig = getattr(chk, meth)
        setattr(chk, meth, (lambda o, k: lambda *a, **kw: (hits.__setitem__(k, hits[k] + 1), o(*a, **kw))[1])(orig, meth))
    for graph in (0, 1, 2):         # 2 = ONE ceiling-sized body with the executed-tile count read from the "device" (core/graphs.py dynamic mode)
        net = build_swiftnet(cfg["backbone"])
        net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()), strict=True)
        net.eval()
        model = blockcopy.BlockCopyModel(net, default_settings(block_policy="all", block_size=cfg["block_size"], block_graph=graph))
        grids = [torch.from_numpy(G[f"grid{t}"]) for t in range(cfg["n_frames"])]
        model.policy = make_forced_policy(cfg["block_size"], grids)
        model = fold_batchnorm(model).to(memory_format=torch.channels_last)
        model.reset_temporal()
        with torch.no_grad():
            for t in range(cfg["n_frames"]):
                y = model(seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"])))
                assert float((y - torch.from_numpy(G[f"logits{t}"])).abs().max()) <= 1e-4, (graph, t)
        assert is_nhwc(y), "the output map should have stayed channels-last"
    # residual block ends go either into the residual gather (library conv route) or into the deferred fused conv's epilogue
    executed = 3 * (cfg["n_frames"] - 1)        # three passes (eager plumbing, graph-mode plumbing, dynamic-mode plumbing); one frame of the clip executes nothing
    assert hits["pad_ring_add"] + hits["conv3x3_ring"] >= 8 * executed and hits["conv3x3_ring"] >= 4 * executed, hits
    assert hits["maxpool3x3s2_ring"] >= executed, hits
    # the output stage (BN -> ReLU -> 1x1 conv to 19 classes -> out-of-place combine) is ONE launch on every executed frame, in the eager
    # engine (TensorWrapper.combine) and in the graph body (slot words); the stand-alone scatter+copy is only called from inside it
    # (graph mode learns the output geometry on its very first frame, which therefore still ends with a stand-alone combine)
    ran = 3 * sum(1 for t in range(cfg["n_frames"]) if G[f"grid{t}"].any())
    assert hits["head1x1_scatter"] == ran - 2 and hits["combine_copy"] == ran - 2, hits      # (two graph modes, each with its first frame)


def test_conv3x3_weight_packing_matches_the_header_formula():
    """pack_conv3x3_weights is the pure permutation include/blockcopy_hip.h documents for bc_conv3x3_ring_nhwc:
    wpk[nb][chunk][tap][cg][lane][j] = W[32*nb + lane%32][32*chunk + 8*cg + 4*(lane//32) + j][tap//3][tap%3]."""
    import torch

    from blockcopy.backend import HipBackend

    Cout, Cin = 64, 96
    w = torch.arange(Cout * Cin * 9, dtype=torch.float32).reshape(Cout, Cin, 3, 3)
    both = HipBackend.pack_conv3x3_weights(w)
    # fp32 3x3: the direct stream (9 values per (cout, cin) pair), the two Winograd F(2x2,3x3) streams (16 each), the F(4x4,3x3) stream (36) and the
    # split stream of the direct form on the 16-bit matrix pipe (9: the direct stream position by position, four weights -> [hi0..3 | lo0..3] fp16)
    # ... and the split stream of the F(4x4) form (36: its stream position by position, [hi0..3 | lo0..3] fp16 of 256 U)
    assert both.numel() == 122 * Cout * Cin
    sp = HipBackend.pack_conv3x3_weights(w / 977.0)
    u4, u8 = sp[41 * Cout * Cin:77 * Cout * Cin].reshape(-1, 4), sp[86 * Cout * Cin:].view(torch.float16).reshape(-1, 8).double()
    assert float((256.0 * u4.double() - (u8[:, :4] + u8[:, 4:])).abs().max()) <= 256.0 * float(u4.abs().max()) * 2.0 ** -21
    assert torch.equal(u8[:, :4].half(), (256.0 * u4).half())
    d4, s8 = sp[:9 * Cout * Cin].reshape(-1, 4), sp[77 * Cout * Cin:86 * Cout * Cin].view(torch.float16).reshape(-1, 8).double()
    assert float((16.0 * d4.double() - (s8[:, :4] + s8[:, 4:])).abs().max()) <= 16.0 * float(d4.abs().max()) * 2.0 ** -21     # hi + lo = 16 w to 22 bits
    assert torch.equal(s8[:, :4].half(), (16.0 * d4).half())
    wpk = both[:9 * Cout * Cin].reshape(Cout // 32, Cin // 32, 9, 4, 64, 4)
    assert wpk.numel() == w.numel() and sorted(wpk.reshape(-1).tolist()) == sorted(w.reshape(-1).tolist())
    # Winograd stream: wino[nb16][chunk][step][q][lane = 16*kq + n][e] = (G g Gt)[f][cin = 32*chunk + 8*step + 2*kq + t][cout = 16*nb16 + n], 2*f + t = 4*q + e
    wr = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    wino = HipBackend.pack_conv3x3_weights(wr.float())[9 * Cout * Cin:25 * Cout * Cin].reshape(Cout // 16, Cin // 32, 4, 8, 64, 4)
    wide = HipBackend.pack_conv3x3_weights(wr.float())[25 * Cout * Cin:41 * Cout * Cin].reshape(Cout // 32, Cin // 32, 8, 8, 64, 4)
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    rngw = np.random.default_rng(1)
    for _ in range(300):
        nb, chunk, step, q, lane, e = (int(rngw.integers(n)) for n in (Cout // 16, Cin // 32, 4, 8, 64, 4))
        f, t = divmod(4 * q + e, 2)
        cin, cout = 32 * chunk + 8 * step + 2 * (lane // 16) + t, 16 * nb + lane % 16
        U = G @ wr.float().double()[cout, cin] @ G.T
        assert abs(float(wino[nb, chunk, step, q, lane, e]) - float(U[f // 4, f % 4])) <= 1e-6
    # wide-tile stream: wino32[nb32][chunk][ss][q][lane = 32*h + n][e] = (G g Gt)[f][cin = 32*chunk + 4*ss + 2*h + t][cout = 32*nb32 + n], 2*f + t = 4*q + e
    for _ in range(300):
        nb, chunk, ss, q, lane, e = (int(rngw.integers(n)) for n in (Cout // 32, Cin // 32, 8, 8, 64, 4))
        f, t = divmod(4 * q + e, 2)
        cin, cout = 32 * chunk + 4 * ss + 2 * (lane // 32) + t, 32 * nb + lane % 32
        U = G @ wr.float().double()[cout, cin] @ G.T
        assert abs(float(wide[nb, chunk, ss, q, lane, e]) - float(U[f // 4, f % 4])) <= 1e-6
    # F(4x4,3x3) stream: wino4[cb][chunk][f][lane = 16*kq + n][j] = (G4 g G4t)[f][cin = 16*chunk + 4*kq + j][cout = 16*cb + n], f = 6*xi + nu
    w4s = HipBackend.pack_conv3x3_weights(wr.float())[41 * Cout * Cin:77 * Cout * Cin].reshape(Cout // 16, Cin // 16, 36, 64, 4)
    G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
    for _ in range(300):
        cb, chunk, f, lane, j = (int(rngw.integers(n)) for n in (Cout // 16, Cin // 16, 36, 64, 4))
        cin, cout = 16 * chunk + 4 * (lane // 16) + j, 16 * cb + lane % 16
        U = G4 @ wr.float().double()[cout, cin] @ G4.T
        assert abs(float(w4s[cb, chunk, f, lane, j]) - float(U[f // 6, f % 6])) <= 1e-6
    rng = np.random.default_rng(0)
    for _ in range(500):
        nb, chunk, tap, cg, lane, j = (int(rng.integers(n)) for n in (Cout // 32, Cin // 32, 9, 4, 64, 4))
        want = w[32 * nb + lane % 32, 32 * chunk + 8 * cg + 4 * (lane // 32) + j, tap // 3, tap % 3]
        assert wpk[nb, chunk, tap, cg, lane, j] == want
    # 16-bit weights: 8 elements per 16-byte vector, 2 steps per 32-channel unit
    w16 = torch.arange(Cout * Cin * 9, dtype=torch.float32).reshape(Cout, Cin, 3, 3).to(torch.int16)      # (exact integers; the permutation is dtype-agnostic)
    wpk16 = HipBackend.pack_conv3x3_weights(w16).reshape(Cout // 32, Cin // 32, 9, 2, 64, 8)
    for _ in range(500):
        nb, unit, tap, step, lane, j = (int(rng.integers(n)) for n in (Cout // 32, Cin // 32, 9, 2, 64, 8))
        assert wpk16[nb, unit, tap, step, lane, j] == w16[32 * nb + lane % 32, 32 * unit + 16 * step + 8 * (lane // 32) + j, tap // 3, tap % 3]


def test_fusion_switch_off_gives_the_same_logits(golden_dir, oracle_backend):
    """BLOCKCOPY_FUSE=0 (fusion.set_enabled(False)): every elementwise op launches separately, dense maps inside
    noblocks modules are plain tensors again, no residual gather / fused pool -- the logits must not move."""
    from blockcopy.core import fusion

    G, cfg = load_golden(golden_dir, "swiftnet_rn18_a.npz")
    prev = fusion.set_enabled(False)
    try:
        errs, _ = run_golden_clip(G, cfg, "cpu", "fused")
    finally:
        fusion.set_enabled(prev)
    assert max(errs) <= 2e-5, errs


def test_pyramid_pooling_fast_path_is_taken_and_changes_nothing(golden_dir, oracle_backend):
    """core/spp_fused.py: inside blockcopy_noblocks the reference-shaped SpatialPyramidPooling runs as spp_levels + spp_fuse (one
    call each per frame, spied on the checker backend), the golden logits hold, and with the switch off (generic op-by-op route)
    they hold too; a module that does not match (another class name, a level grid option) takes the generic route."""
    import blockcopy.backend as bk
    from blockcopy.core import fusion, spp_fused
    from bc_workloads.swiftnet import SpatialPyramidPooling

    be = bk.get_backend()
    calls = []
    o_lv, o_fu = be.spp_levels, be.spp_fuse
    be.spp_levels = lambda *a, **k: (calls.append("levels"), o_lv(*a, **k))[1]
    be.spp_fuse = lambda *a, **k: (calls.append("fuse"), o_fu(*a, **k))[1]
    import blockcopy
    from blockcopy.core.argparser import default_settings
    from bc_workloads import seeded
    from bc_workloads.bn_fold import fold_batchnorm
    from bc_workloads.swiftnet import build_swiftnet

    G, cfg = load_golden(golden_dir, "swiftnet_rn18_a.npz")

    def clip_errors(channels_last):
        net = build_swiftnet(cfg["backbone"])
        net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()), strict=True)
        model = blockcopy.BlockCopyModel(net.eval(), default_settings(block_policy="all", block_size=cfg["block_size"]))
        model.policy = make_forced_policy(cfg["block_size"], [torch.from_numpy(G[f"grid{t}"]) for t in range(cfg["n_frames"])])
        model = fold_batchnorm(model)
        if channels_last:
            model = model.to(memory_format=torch.channels_last)
        model.reset_temporal()
        with torch.no_grad():
            return [float((model(seeded.synthetic_frame(cfg["frame_seed0"] + t, (cfg["N"], 3, cfg["H"], cfg["W"]))) - torch.from_numpy(G[f"logits{t}"])).abs().max())
                    for t in range(cfg["n_frames"])]

    try:
        errs = clip_errors(True)
        n_frames = len(errs)
        # (one frame of this clip executes no tile at all and returns the cached output: no network pass)
        assert max(errs) <= 1e-4 and calls.count("levels") == calls.count("fuse") and n_frames - 1 <= calls.count("fuse") <= n_frames, (errs, calls)
        del calls[:]
        assert max(clip_errors(False)) <= 1e-4 and not calls            # an NCHW model keeps NCHW maps: generic route
        fusion.SPP_FUSED = False
        assert max(clip_errors(True)) <= 1e-4 and not calls
    finally:
        fusion.SPP_FUSED = True
        be.spp_levels, be.spp_fuse = o_lv, o_fu
    spp = SpatialPyramidPooling(64, 3, bt_size=32, level_size=8, out_size=64, grids=(4, 2, 1)).eval()
    assert spp_fused.match(spp) is not None
    spp.square_grid = True
    assert spp_fused.match(spp) is None
    spp.square_grid = False
    spp.train()
    assert spp_fused.match(spp) is None
    spp.eval()
    spp.upsampling_method = lambda x, size: torch.nn.functional.interpolate(x, size, mode="nearest")
    assert spp_fused.match(spp) is None


def test_spp_support_check_mirrors_the_launchers_lds_budget():
    """HipBackend.spp_supported (pure shape logic, no library needed) refuses what bc_spp_levels_nhwc / bc_spp_fuse_nhwc refuse: the
    reference's own SpatialPyramidPooling defaults (bt_size 512, level_size 128 -> 268 KB of LDS) must take the generic route, not raise
    BC_ERR_SHAPE in the middle of a frame (round-3 advisor)."""
    from blockcopy.backend import HipBackend

    x128 = torch.empty((1, 32, 64, 128), dtype=torch.float32).permute(0, 3, 1, 2)        # SwiftNet-RN18: 128 channels, level size 42
    assert HipBackend.spp_supported(x128, 42, 3, 128, [(8, 16), (4, 8), (2, 4)])
    x512 = torch.empty((1, 32, 64, 512), dtype=torch.float32).permute(0, 3, 1, 2)
    assert not HipBackend.spp_supported(x512, 128, 3, 128, [(8, 16), (4, 8), (2, 4)])      # levels: (3 * 512 + 512 * 128) * 4 = 268 KB
    assert HipBackend.spp_supported(x128, 42, 3, 128)                                     # without grids: the levels budget only
    fine = [(32, 64), (16, 32), (8, 16)]                                                   # 2688 bins x 42 channels of level maps: 451 KB
    assert not HipBackend.spp_supported(x128, 42, 3, 128, fine)


def test_c1_cpu_plumbing_4x512x1024_block128_all_active(oracle_backend, golden_dir):
    """BASELINE config C1 at its stated shape: SwiftNet-RN18 on 4 synthetic 512x1024 frames, block 128 (4x8 tiles),
    policy forced 100 %-active, no GPU (the block ops are served by the checker backend).  The logits equal the REFERENCE's for
    the same clip (tests/golden/swiftnet_rn18_c1.npz, <= 2e-5), both engines agree, every tile is executed on every frame, the
    packed encoder equals the dense one (P1) and frame_state is the input."""
    import blockcopy
    from blockcopy.core import tensorwrapper as tw
    from bc_workloads import harness, seeded

    frames = [seeded.synthetic_frame(t, (1, 3, 512, 1024)) for t in range(4)]   # torch.randn, seeds 0..3
    outs = {}
    try:
        for engine in ("fused", "reference"):
            tw.set_engine(engine)
            model = harness.build_model("resnet18", block_policy="all", block_size=128, device="cpu")
            model.reset_temporal()
            with torch.no_grad():
                ys = []
                for f in frames if engine == "fused" else frames[:2]:
                    ys.append(model(f).clone())
                    assert model.policy_meta["num_exec"] == 32 and model.policy_meta["num_total"] == 32
                    assert torch.equal(model.policy_meta["frame_state"], f)
            outs[engine] = ys
            assert ys[0].shape == (1, 19, 128, 256) and all(torch.isfinite(y).all() for y in ys)
    finally:
        tw.set_engine("fused")
    for a, b in zip(outs["fused"], outs["reference"]):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    G, cfg = load_golden(golden_dir, "swiftnet_rn18_c1.npz")
    assert (cfg["H"], cfg["W"], cfg["block_size"], cfg["frame_seed0"], cfg["n_frames"]) == (512, 1024, 128, 0, 4)
    for t, y in enumerate(outs["fused"]):
        assert golden_logit_error(G, cfg, t, y) <= 2e-5 * max(1.0, float(G[f"logits{t}_absmax"])), t
    dense = harness.build_model("resnet18", block_policy="static", device="cpu")
    model = harness.build_model("resnet18", block_policy="all", block_size=128, device="cpu")
    with torch.no_grad():
        want = dense.forward_down(frames[0])
        xw = blockcopy.to_tensorwrapper(frames[0])
        xw.process_temporal_features(None)
        grid = torch.ones(1, 1, 4, 8, dtype=torch.bool)
        got = model.base_model.forward_down(xw.to_blocks(grid, grid))
        for w, g in zip(want, got):
            assert float((g.combine().to_tensor() - w).abs().max()) <= 1e-4 * max(1.0, float(w.abs().max()))


def test_derived_parameter_cache_evicts_the_dead_before_the_living():
    """core/fusion.py keeps derived per-parameter tensors (folded BN vectors, packed conv weights).  When the cache fills up -- an online
    policy leaves one entry per optimizer step -- the entries of freed tensors and of superseded versions go, those of live, unchanged
    parameters stay: a model that was warmed up must not have to re-derive its packed weights inside a later graph capture."""
    from blockcopy.core import fusion

    fusion.clear_cache()
    live = torch.nn.Parameter(torch.arange(8.0))
    v_live = fusion.channel_vector(live)
    trained = torch.nn.Parameter(torch.zeros(4))
    for step in range(4200):                      # (in-place updates: a new version, hence a new entry, per step)
        with torch.no_grad():
            trained.add_(1.0)
        fusion.channel_vector(trained)
        if step % 7 == 0:
            fusion.channel_vector(torch.nn.Parameter(torch.ones(3)))      # (dies at once)
    assert len(fusion._cache) <= 4097
    assert fusion.channel_vector(live) is v_live, "the live parameter's entry was dropped"
    assert float(fusion.channel_vector(trained)[0]) == 4200.0
    fusion.clear_cache()
