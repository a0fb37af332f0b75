#!/usr/bin/env python3
"""C3: what a frame's policy part costs -- per-frame wall time of policy.forward / model / policy.optim with a device synchronisation
around each (a measurement aid, not the bench), and whether the no-grad frames run the captured trunk.  usage: python tools/c3_policy_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
import torch
from bc_workloads import harness

torch.manual_seed(20260)
model = harness.build_model("resnet18", block_policy="rl_semseg", block_size=128, block_target=0.3, device="cuda", dtype=torch.float32, seed=0,
                            block_graph=1, block_train_interval=3, channels_last=True)
clips = [harness.synthetic_clip(20, (1, 3, 1024, 2048), seed=s, device="cuda") for s in range(2)]
pol = model.policy
orig_fwd, orig_opt = pol.forward, pol.optim
acc = {"fwd_train": [], "fwd_nograd": [], "optim_train": [], "optim_nograd": []}


def fwd(pm):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = orig_fwd(pm)
    torch.cuda.synchronize()
    if pm.get("outputs") is not None:
        acc["fwd_train" if pm.get("train_hint", True) else "fwd_nograd"].append(time.perf_counter() - t)
    return r


def opt(pm, train=True):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = orig_opt(pm, train=train)
    torch.cuda.synchronize()
    acc["optim_train" if train else "optim_nograd"].append(time.perf_counter() - t)
    return r


pol.forward, pol.optim = fwd, opt
with torch.no_grad():
    for rep in range(3):
        for clip in clips:
            harness.run_clip(model, clip)
torch.cuda.synchronize()
for k, v in acc.items():
    v = v[len(v) // 3:]
    print(f"{k}: {len(v)} samples, median {sorted(v)[len(v) // 2] * 1e3:.3f} ms" if v else f"{k}: none")
print("captured trunks:", {k[0]: (st['graph'] is not None, st['warm']) for k, st in pol._fwd_graphs.items()})
