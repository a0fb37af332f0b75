"""Policy net on own kernels at a config's geometry: forward / REINFORCE step time (eager launches and hipGraph replay) next to the
autograd route on the same module.  python tools/kbench_pnet.py [--batch 1] [--height 1024 --width 2048 --block 128]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"), ROOT):
    sys.path.insert(0, p)

import torch
import torch.nn.functional as F


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, (t1 - t0) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--block", type=int, default=128)
    ap.add_argument("--ops", type=int, default=1, help="per-op timing of the two sequences (eager, events around each op)")
    args = ap.parse_args()
    from blockcopy.core.argparser import default_settings
    from blockcopy.policy import native
    from blockcopy.policy.policy import build_policy_from_settings
    from blockcopy.policy.information_gain import InformationGainSemSeg

    torch.manual_seed(0)
    N, H, W = args.batch, args.height, args.width
    pol = build_policy_from_settings(default_settings(block_policy="rl_semseg", block_size=args.block, block_target=0.3)).cuda()
    pol.net.train().to(memory_format=torch.channels_last)
    nat = native.NativePolicyNet(pol.net, pol.optimizer, (N, 3, H, W), "cuda")
    h, w = nat.h, nat.w
    x = torch.randn((N, 26, h, w), device="cuda").contiguous(memory_format=torch.channels_last)
    feat = torch.zeros((N, h, w, 32), device="cuda")
    feat[..., :26] = x.permute(0, 2, 3, 1)
    outputs = torch.randn((N, 19, H // 4, W // 4), device="cuda").contiguous(memory_format=torch.channels_last)
    prev = outputs + 0.3 * torch.randn_like(outputs)
    grid = torch.rand((N, 1, nat.GH, nat.GW), device="cuda") > 0.5
    ig_mod = InformationGainSemSeg(19)

    def torch_fwd():
        with torch.no_grad():
            return pol.net.layers(pol.net.backbone(x))

    def torch_step():
        logits = pol.net.layers(pol.net.backbone(x))
        ig = ig_mod({"outputs": outputs, "outputs_prev": prev})
        reward = F.adaptive_max_pool2d(ig + 0.1, output_size=grid.shape[2:])
        reward = torch.where(grid, reward, -reward)
        loss = (F.binary_cross_entropy_with_logits(logits, grid.float(), reduction="none") * reward.detach()).mean()
        loss.backward()
        pol.optimizer.step()
        pol.optimizer.zero_grad(set_to_none=True)

    rows = {}
    rows["torch forward (no_grad, eager)"] = timed(torch_fwd)
    rows["torch forward + backward + RMSprop (eager)"] = timed(torch_step, reps=10)
    native.USE_GRAPH = False
    rows["native forward (eager launches)"] = timed(lambda: nat.forward_on(feat))
    rows["native step (eager launches)"] = timed(lambda: nat.step(grid, outputs, prev, 0.35, 0.3, 5.0), reps=10)
    native.USE_GRAPH = True
    rows["native forward (hipGraph)"] = timed(lambda: nat.forward_on(feat))
    rows["native step (hipGraph)"] = timed(lambda: nat.step(grid, outputs, prev, 0.35, 0.3, 5.0), reps=10)
    print(f"policy net N={N} input {h}x{w}: {len(nat._fwd_ops)} forward ops, {len(nat._step_ops)} step ops")
    for k, (gpu, host) in rows.items():
        print(f"  {k:48s} {gpu:8.3f} ms GPU   {host:8.3f} ms host")
    if args.ops:
        native.USE_GRAPH = False
        for name, ops in (("forward", nat._fwd_ops), ("step", nat._step_ops)):
            st = torch.cuda.current_stream().cuda_stream
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
            for _ in range(2):
                evs[0].record()
                for i, f in enumerate(ops):
                    f(st)
                    evs[i + 1].record()
            torch.cuda.synchronize()
            print(f"  -- {name}: per op (us)")
            print("    " + " ".join(f"{evs[i].elapsed_time(evs[i + 1]) * 1e3:.0f}" for i in range(len(ops))))


if __name__ == "__main__":
    main()
