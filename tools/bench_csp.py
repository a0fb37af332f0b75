#!/usr/bin/env python3
"""Config C5: CSP-ResNet50 pedestrian detector on synthetic 1024x2048 clips, block 128, fixed 30 % mask (the reference's
target, configs/elephant/cityperson/csp_r50_clip_blockcopy_030.py:11), one independent video stream per GPU.

    python tools/bench_csp.py [--steps K] [--warmup W] [--half]          (N GPUs: torch.distributed.run, as bench.py)

Prints one JSON line: frames/s of the block path (whole job), the dense detector on the same GPU, and the large
scatter+copy launches of the head (3 x (1,256,256,512) maps per frame)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--half", action="store_true")
    ap.add_argument("--target", type=float, default=0.3)
    ap.add_argument("--channels-last", type=int, default=1)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--clip-len", type=int, default=20)
    args = ap.parse_args()

    import blockcopy.backend as bk
    from bc_workloads import harness, replicas
    from bc_workloads.csp import build_csp

    rank, world, local = replicas.dist_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=device)
    torch.backends.cudnn.benchmark = True
    dtype = torch.float16 if args.half else torch.float32
    be = bk.get_backend()
    model = build_csp(block_policy="fixed", block_size=128, block_target=args.target, device=device, dtype=dtype,
                      channels_last=bool(args.channels_last), block_graph=args.graph, seed=1000 * rank)
    clip = harness.synthetic_clip(args.clip_len, (1, 3, 1024, 2048), seed=rank * 100, device=device, dtype=dtype)

    def run_clip(m):
        if hasattr(m, "reset_temporal"):
            m.reset_temporal()
        for f in clip:
            m.simple_test(f)

    for _ in range(args.warmup):
        run_clip(model)
    be.prof_reset()
    be.prof_enable(["combine_copy"])
    replicas.barrier(world, device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_clip(model)
    replicas.barrier(world, device)
    elapsed = time.perf_counter() - t0
    be.prof_enable([])
    cc = be.prof_read("combine_copy")
    fps, elapsed, frames = replicas.job_throughput(args.steps * args.clip_len, elapsed, world, device)
    if rank == 0:
        out = {"config": f"C5: CSP-ResNet50 1x3x1024x2048 clips of {args.clip_len}, block 128, fixed {args.target:.0%} mask, "
                         f"{'channels-last' if args.channels_last else 'NCHW'}{', hipGraph' if args.graph else ''}, {world} independent stream(s)",
               "value": fps, "unit": "frames/s", "n_gpus": world, "dtype": "f16" if args.half else "f32",
               "exec_fraction": model.policy.stats.get_exec_percentage()}
        if cc["launches"]:
            out["scatter_copy"] = {"launches": cc["launches"], "avg_us": 1e3 * cc["total_ms"] / cc["launches"],
                                   "GBps": cc["total_bytes"] / (cc["total_ms"] * 1e-3) / 1e9,
                                   "MB_per_launch": cc["total_bytes"] / cc["launches"] / 1e6}
        if world == 1:
            dense = build_csp(block_policy="static", device=device, dtype=dtype, channels_last=bool(args.channels_last))
            run_clip(dense)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            run_clip(dense)
            torch.cuda.synchronize(device)
            out["dense_gpu_fps"] = args.clip_len / (time.perf_counter() - t0)
            out["speedup_vs_dense_gpu"] = fps / out["dense_gpu_fps"]
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
