"""Where the HOST spends a C3 frame: cProfile over a few clips of the online-RL loop (rl_semseg, target 0.3, train-interval 3) at
1024x2048, plus wall / fps of the same loop for both graph modes.  python tools/c3_host_profile.py [--graph 1|2] [--profile 1]"""
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
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--profile", type=int, default=1)
    ap.add_argument("--clips", type=int, default=6)
    ap.add_argument("--half", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--target", type=float, default=0.3)
    args = ap.parse_args()
    from bc_workloads import harness

    torch.manual_seed(20260)
    random.seed(20260)
    dtype = torch.float16 if args.half else torch.float32
    model = harness.build_model("resnet18", block_policy="rl_semseg", block_size=128, block_target=args.target, device="cuda", dtype=dtype, channels_last=True,
                                block_graph=args.graph, block_train_interval=3)
    shape = (args.batch, 3, 1024, 2048)
    clips = [harness.synthetic_clip(20, shape, seed=100 * c, device="cuda", dtype=dtype) for c in range(2)]
    harness.run_clip(model, clips[0][:1])
    model.prewarm(clips[0][0])
    for i in range(3):
        harness.run_clip(model, clips[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.clips):
        harness.run_clip(model, clips[i % 2])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    n = args.clips * 20 * args.batch
    print(f"graph={args.graph} half={args.half} batch={args.batch}: {n / t1:.1f} fps, {1e3 * t1 / n:.3f} ms/frame wall, host enqueue {1e3 * t_host / n:.3f} ms/frame, "
          f"executed {model.policy.stats.get_exec_percentage():.3f}")
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        for i in range(3):
            harness.run_clip(model, clips[i % 2])
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr, stream=sys.stdout)
        st.strip_dirs()
        st.sort_stats("tottime").print_stats(40)
        st.sort_stats("cumulative").print_stats(60)


if __name__ == "__main__":
    main()
