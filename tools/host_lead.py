#!/usr/bin/env python3
"""How far the host runs ahead of the GPU in the captured frame loop (config C2 by default): per frame, the host time at which the frame's last
command was enqueued against the GPU time at which the frame ended (events), the time the host spends blocked, and the cProfile top of the
loop.  usage: python tools/host_lead.py [--policy fixed] [--target 0.5] [--ring-depth N]"""
import argparse
import cProfile
import os
import pstats
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--policy", default="fixed")
    ap.add_argument("--target", type=float, default=0.5)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--clips", type=int, default=6)
    a = ap.parse_args()
    from bc_workloads import harness

    torch.manual_seed(20260)
    random.seed(20260)
    model = harness.build_model("resnet18", block_policy=a.policy, block_size=128, block_target=a.target, device="cuda", dtype=torch.float32, channels_last=True,
                                block_graph=a.graph, block_train_interval=3)
    clips = [harness.synthetic_clip(20, (1, 3, 1024, 2048), seed=100 * c, device="cuda", dtype=torch.float32) for c in range(2)]
    harness.run_clip(model, clips[0][:1])
    if hasattr(model, "prewarm"):
        model.prewarm(clips[0][0])
    for i in range(3):
        harness.run_clip(model, clips[i % 2])
    torch.cuda.synchronize()
    # per-frame: host enqueue-done time vs GPU frame-end time
    t_host, evs = [], []
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    t0 = time.perf_counter()
    for c in range(2):
        model.reset_temporal()
        for f in clips[c]:
            model(f)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            evs.append(ev)
            t_host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t_gpu = [base.elapsed_time(e) * 1e-3 for e in evs]
    lead = [g - h for g, h in zip(t_gpu, t_host)]
    print("frame: host enqueue done [ms] | GPU frame end [ms] | GPU end - host done [ms] (positive: the host is ahead)")
    for i in list(range(0, 8)) + list(range(32, 40)):
        print(f"  {i:2d}: {1e3 * t_host[i]:8.3f} | {1e3 * t_gpu[i]:8.3f} | {1e3 * lead[i]:7.3f}")
    print(f"wall {1e3 * t_gpu[-1] / len(evs):.3f} ms/frame; mean lead {1e3 * sum(lead[5:]) / len(lead[5:]):.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(3):
        harness.run_clip(model, clips[i % 2])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.strip_dirs()
    st.sort_stats("tottime").print_stats(12)


if __name__ == "__main__":
    main()
