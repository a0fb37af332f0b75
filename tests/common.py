"""Helpers shared by CPU and GPU tests."""
import json
import os

import numpy as np
import torch


def load_golden(golden_dir, name):
    G = np.load(os.path.join(golden_dir, name))
    cfg = json.loads(bytes(G["cfg"]).decode()) if "cfg" in G.files else None
    return G, cfg


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
                errs.append(float((y.cpu() - torch.from_numpy(G[f"logits{t}"])).abs().max()))
                if f"frame_state{t}" in G.files:
                    fs_errs.append(float((model.policy_meta["frame_state"].cpu() - torch.from_numpy(G[f"frame_state{t}"])).abs().max()))
    return errs, fs_errs
