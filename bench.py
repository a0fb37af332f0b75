#!/usr/bin/env python3
"""Headline benchmark: frames/s of SwiftNet-RN18 through the block-copy engine on synthetic 1024x2048 clips at a fixed
50 % execution mask (BASELINE.json config C2), plus the HBM roofline of the fused scatter+copy kernel and the host-CPU
dense baseline.

    python bench.py --gpus N --steps K --warmup W

N > 1 run directly: bench.py starts its own N worker processes (one per GPU) before anything touches the GPU; under
``torch.distributed.run`` (WORLD_SIZE already set) it is a worker.  Every worker pins itself to its own slice of the host
cores and gets its own MIOpen user-db / cache directory before torch is imported; all workers read the same conv plan
table (blockcopy/plans/gfx950.json), so no rank re-tunes and all ranks run the same kernel forms.

The printed JSON line stays small (< 4 KB); the per-layer plan list, the PMC traffic table and the per-rank numbers go
to gpurun_out/bench_details_<config>.json.

A *step* is one 20-frame clip (reset_temporal(); frame 0 executes all 128 tiles, frames 1..19 execute a seeded 64 of
128) with every frame already resident in HBM.  fps = frames / wall, device-synchronised on both sides, exactly as the
reference's driver measures it (semantic_segmentation/test_swiftnet.py:147-172).  Clips are independent, so N GPUs run
N replicas with no collective on the data path (the barrier/all-reduce below only brackets the clock); scaling is weak.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"),):
    if _p not in sys.path:
        sys.path.insert(0, _p)

torch = None   # imported by the worker only: the launcher parent never touches torch or the GPU

METRIC = "frames/sec SwiftNet-RN18 1024×2048 @ 50% active blocks, 1→8 GPU; scatter GB/s"
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
CLIP_LEN = 20


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5, help="timed clips per rank")
    ap.add_argument("--warmup", type=int, default=2, help="untimed clips per rank")
    ap.add_argument("--half", action="store_true", help="fp16 compute (reference speed configs use --half); default fp32")
    ap.add_argument("--backbone", default="resnet18")
    ap.add_argument("--batch", type=int, default=1, help="clips processed side by side (the reference's speed configs use 2); C2 is defined at 1")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--block-size", type=int, default=128)
    ap.add_argument("--target", type=float, default=0.5)
    ap.add_argument("--policy", default="fixed", choices=["fixed", "random", "all", "rl_semseg"],
                    help="fixed = seeded fixed-fraction mask (config C2, the headline); rl_semseg = online-trained policy (C3)")
    ap.add_argument("--train-interval", type=int, default=3)
    ap.add_argument("--channels-last", type=int, default=1,
                    help="1 (default): channels-last weights/activations -> NHWC packed tiles, MIOpen's preferred layout; 0: NCHW")
    ap.add_argument("--also-half", type=int, default=1,
                    help="1 (default, N=1 only): after the timed fp32 region also measure the same workload in fp16 "
                         "(the reference's speed configs use --half) and report it under kernels.fp16")
    ap.add_argument("--timings", type=int, default=0, help="profiler section level (blockcopy.utils.profiler); report goes to stderr")
    ap.add_argument("--engine", default="fused", choices=["fused", "reference"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-GPU comparison run")
    ap.add_argument("--miopen-find", type=int, default=1, help="torch.backends.cudnn.benchmark during warm-up")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--graph", type=int, default=None,
                    help="hipGraph replay of the packed pipeline: 0 = eager launches, 1 = one graph per executed-tile count, 2 = ONE graph for "
                         "every count (launches sized for all tiles, count read from the device; no wait for a device-side policy decision). "
                         "Default: 2 for the online-RL policies in fp16, 1 otherwise")
    ap.add_argument("--config", default=None, choices=["C2", "C3", "C3h", "C4", "C5"],
                    help="BASELINE.json config preset: C2 (default workload), C3 = --policy rl_semseg --target 0.3, C3h = the reference's own speed "
                         "scenario (configs/swiftnet_rn18/swiftnet_rn18_rl05_speed.sh: rl_semseg, target 0.5, --half, batch 2, train-interval 3), "
                         "C4 = SwiftNet-RN50 2048x4096 block 64 target 0.25, C5 = CSP-ResNet50 pedestrian detector 1024x2048 block 128 target 0.3 "
                         "(one stream per GPU)")
    ap.add_argument("--upload-variant", type=int, default=1, help="1 (default, N=1): also report the PCIe-inclusive fps of the reference's full loop")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="(tests only) replace the GPU workload by a tiny CPU stand-in so that the launcher, the per-rank "
                         "environment and the clock bracket can be exercised on a machine without a GPU")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="(tests only) let several replicas share a GPU when fewer than --gpus devices are visible")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="seconds before the launcher gives up on its replicas")
    ap.add_argument("--save-plan", default=None, metavar="FILE",
                    help="after the run write the conv plan table (loaded + measured entries) to FILE (tools/tune_plans.py)")
    ap.add_argument("--details", default=None, metavar="FILE", help="side file for the long tables (default gpurun_out/bench_details_<config>.json)")
    args = ap.parse_args(argv)
    args.workload = "swiftnet"
    if args.config == "C3":
        args.policy, args.target = "rl_semseg", 0.3
    elif args.config == "C3h":
        args.policy, args.target, args.half, args.batch, args.train_interval = "rl_semseg", 0.5, True, 2, 3
    elif args.config == "C4":
        args.backbone, args.height, args.width, args.block_size, args.target = "resnet50", 2048, 4096, 64, 0.25
    elif args.config == "C5":
        args.workload, args.target, args.backbone = "csp", 0.3, "csp_resnet50"
    if args.graph is None:
        # measured (profiles/r04): with an online-RL policy the ONE dynamic graph (no wait for the device-side decision) wins where the
        # frame is short -- fp16, batch 2: C3h 882-905 vs 819 fps -- and loses ~10 % to the sixteen per-count graphs in fp32 batch 1
        # (C3 499-508 vs 560-562), whose conv plans are tuned per count while the dynamic graph runs one plan for every count
        args.graph = 2 if (args.policy.startswith("rl_") and args.workload != "csp" and args.channels_last and args.half) else 1
    return args


def cpu_dense_baseline(args, n_frames):
    """'Reference on the host CPU cores, block execution disabled': dense SwiftNet fp32 through PyTorch CPU convs
    (oneDNN) on all host cores, BN folded, no_grad; bounded sample."""
    from bc_workloads import harness, seeded

    cores = torch.get_num_threads()
    model = build_workload(args, "static", torch.float32, "cpu", 0)
    if args.workload == "csp":
        model = model.det.head_maps       # backbone + head maps on the host (the box decode / NMS of this package exists as HIP only)
    x = seeded.synthetic_frame(0, (1, 3, args.height, args.width))   # the CPU baseline is always one frame at a time
    with torch.no_grad():
        model(x)   # warm-up (oneDNN primitive creation)
        t0 = time.perf_counter()
        done = 0
        for _ in range(n_frames):
            model(x)
            done += 1
            if time.perf_counter() - t0 > 30.0:
                break
        dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{done} dense (block-exec disabled) {'csp_resnet50 detector (network only, no box decode)' if args.workload == 'csp' else args.backbone} frames 1x3x{args.height}x{args.width} fp32, "
                      f"PyTorch CPU oneDNN conv, BN folded, after 1 warm-up frame ({dt:.1f} s)"}


class _Detector:
    """The CSP detector behind the call convention the clip harness uses (model(frame), reset_temporal(), .policy)."""

    def __init__(self, det):
        self.det = det
        self.policy = getattr(det, "policy", None)

    def reset_temporal(self):
        if hasattr(self.det, "reset_temporal"):
            self.det.reset_temporal()

    def __call__(self, frame):
        return self.det.simple_test(frame)


def build_workload(args, policy, dtype, device, rank, graph=None):
    """The model of the selected config: SwiftNet behind BlockCopyModel (C2-C4) or the CSP detector (C5); policy 'static' = dense."""
    from bc_workloads import harness

    graph = args.graph if graph is None else graph
    # the online-RL policy draws its initial weights and its sampling seed from torch's generator: fixed per (rank, policy), so that a
    # config's executed fraction -- and with it its frames/s -- is the same from run to run
    import random

    import torch as _torch          # (the module-level name is only bound inside a worker process; tools import this function directly)

    _torch.manual_seed(20260 + 1000 * rank)
    random.seed(20260 + 1000 * rank)
    if args.workload == "csp":
        from bc_workloads.csp import build_csp

        kw = {} if policy == "static" else dict(block_size=args.block_size, block_target=args.target, block_graph=graph, seed=1000 * rank)
        return _Detector(build_csp(block_policy=policy, device=device, dtype=dtype, channels_last=bool(args.channels_last), **kw))
    if policy == "static":
        return harness.build_model(args.backbone, block_policy="static", device=device, dtype=dtype, channels_last=bool(args.channels_last))
    return harness.build_model(args.backbone, block_policy=policy, block_size=args.block_size, block_target=args.target, device=device,
                               dtype=dtype, seed=1000 * rank, block_graph=graph, block_train_interval=args.train_interval,
                               channels_last=bool(args.channels_last))


def config_name(args):
    """Which BASELINE.json config the arguments correspond to."""
    if args.workload == "csp":
        return "C5"
    if (args.backbone, args.height, args.width, args.block_size, args.policy, args.target, args.batch, bool(args.half)) == ("resnet18", 1024, 2048, 128, "rl_semseg", 0.5, 2, True):
        return "C3h (reference speed scenario: rl_semseg target 0.5, fp16, batch 2, train-interval 3)"
    if args.batch != 1:
        return f"custom(batch {args.batch})"
    base = (args.backbone, args.height, args.width, args.block_size)
    if base == ("resnet18", 1024, 2048, 128):
        return "C2" if args.policy == "fixed" and args.target == 0.5 else ("C3" if args.policy == "rl_semseg" else "C2-variant")
    if base == ("resnet50", 2048, 4096, 64) and args.policy == "fixed" and args.target == 0.25:
        return "C4"
    return "custom"


def scatter_copy_large(be, device, iters=20):
    """The same fused scatter+copy kernel at the largest map of the configs (C5 detector head, (1,256,256,512) fp32,
    block 32, 64 of 128 tiles = 268 MB of traffic, beyond the 256 MiB Infinity Cache): the HBM-bound figure."""
    N, C, H, W, bs, n_exec = 1, 256, 256, 512, 32, 64
    GH, GW = H // bs, W // bs
    g = torch.zeros(GH * GW, dtype=torch.bool)
    g[torch.randperm(GH * GW, generator=torch.Generator().manual_seed(0))[:n_exec]] = True
    gi = torch.where(g, g.cumsum(0) - 1, (~g).cumsum(0) - 1 - GH * GW).to(torch.int32).view(N, 1, GH, GW).to(device)
    blocks = torch.randn((n_exec, C, bs, bs), device=device)
    prev = torch.randn((N, C, H, W), device=device)
    out = torch.empty_like(prev)
    for _ in range(3):
        be.combine_copy(blocks, prev, out, gi)
    be.prof_reset()
    be.prof_enable(["combine_copy"])
    for _ in range(iters):
        be.combine_copy(blocks, prev, out, gi)
    torch.cuda.synchronize(device)
    be.prof_enable([])
    r = be.prof_read("combine_copy")
    gbps = r["total_bytes"] / (r["total_ms"] * 1e-3) / 1e9
    return {"kernel": "k_combine_copy", "shape": "(1,256,256,512) f32, block 32, 64/128 tiles executed", "launches": r["launches"],
            "avg_launch_us": 1e3 * r["total_ms"] / r["launches"], "algorithmic_bytes_per_launch": r["total_bytes"] / r["launches"],
            "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS}


def pmc_traffic():
    """HBM traffic per launch of the roofline kernel from a separate rocprofv3 --pmc run of this same command
    (tools/pmc_traffic.py writes profiles/traffic_latest.json); None when no such measurement is committed."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(path):
        return (None, None), None, {}
    with open(path) as f:
        t = json.load(f)
    per_kernel = {label: {"kernel": v.get("kernel"), "hbm_MB": round(v["hbm_bytes_per_launch"] / 1e6, 2),
                          "algorithmic_MB": round(v["algorithmic_bytes_per_launch"] / 1e6, 2),
                          "traffic_over_algorithmic": round(v["traffic_over_algorithmic"], 3)}
                  for label, v in t.get("kernels", {}).items() if "traffic_over_algorithmic" in v}
    return (t.get("k_combine_copy_bytes_per_launch"), t.get("k_head1x1_bytes_per_launch")), t.get("source"), per_kernel


def rocprof_cross_check(kernel_prefix):
    """(avg launch us, frames in the window) of the roofline kernel in the COMMITTED rocprofv3 kernel trace of this command's graph replays
    (profiles/rocprof_latest.json, written by tools/trace_summary.py from a separate profiled run), or (None, None)."""
    path = os.path.join(ROOT, "profiles", "rocprof_latest.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        t = json.load(f)
    for name, v in t.get("kernels", {}).items():
        if name.startswith(kernel_prefix):
            return v.get("avg_us"), t.get("window_frames")
    return None, None


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_replicas(args, argv):
    """``python bench.py --gpus N`` (N > 1) run directly: start one worker process per GPU and relay rank 0's JSON line.

    The launcher never imports torch and never touches the GPU (a process that has initialised HIP must not exec or
    fork workers); the workers are plain children (``subprocess``), each with the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT) and its GPU = LOCAL_RANK.  Exit code is
    non-zero if any worker fails; the others are then terminated (only processes started here are signalled).
    The reference has no multi-GPU launcher (semantic_segmentation/test_swiftnet.py:57 pins one device)."""
    import signal
    import subprocess

    world = args.gpus
    port = _free_port()
    procs = []

    def die_with_parent():
        # child side, between fork and exec (nothing has touched the GPU in either process): if the launcher is killed outright
        # (SIGKILL by a driver timeout), the kernel sends the worker SIGTERM instead of leaving it holding a GPU in its own session
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM, 0, 0, 0)      # PR_SET_PDEATHSIG
        except Exception:
            pass

    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BC_BENCH_WORKER="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True,
                                      preexec_fn=die_with_parent))

    def stop_workers(signum, frame):
        raise KeyboardInterrupt(f"signal {signum}")

    for sig in (signal.SIGTERM, signal.SIGHUP):      # a driver timeout stops the launcher with SIGTERM: take the workers down too
        signal.signal(sig, stop_workers)
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)   # drain rank 0's pipe
    reader.start()
    deadline = time.time() + args.launch_timeout
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                raise RuntimeError("replica(s) failed: " + ", ".join(f"rank {r} exit code {c}" for r, c in bad))
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                raise RuntimeError(f"replicas still running after {args.launch_timeout:.0f} s")
            time.sleep(0.1)
    except (RuntimeError, KeyboardInterrupt) as e:
        print(f"bench.py launcher: {e}; stopping the other replicas", file=sys.stderr)
        rc = 1
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)   # the worker's own session: nothing else lives in that group
                except ProcessLookupError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
    reader.join(10)
    out0 = b"".join(c for c in chunks if c)
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    if rc == 0 and not any(line.lstrip().startswith("{") for line in out0.decode(errors="replace").splitlines()):
        print("bench.py launcher: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


def isolate_rank(rank, local, local_world):
    """Per-rank host setup, BEFORE torch / MIOpen are loaded: a disjoint slice of the host cores (N Python processes that each
    enqueue ~1000 launches per ms-frame must not migrate over each other) and a private MIOpen user database / kernel cache
    (eight processes opening the same sqlite files at start-up serialise on its lock, and a find-db written by one rank mid-run
    must not change another rank's solver choice).  Returns what was set, for the details file."""
    info = {}
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = max(1, len(cores) // max(1, local_world))
        mine = cores[local * per:(local + 1) * per] if local_world > 1 else cores
        if mine and local_world > 1:
            os.sched_setaffinity(0, mine)
        info["cpu_cores"] = len(mine)
        info["cpu_first"] = mine[0] if mine else None
        os.environ.setdefault("OMP_NUM_THREADS", str(len(mine)))
    except (AttributeError, OSError):
        pass
    if local_world > 1:
        base = miopen_base_dir()
        for var, sub in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
            if var not in os.environ:
                d = os.path.join(base, f"rank{rank}", sub)
                os.makedirs(d, exist_ok=True)
                os.environ[var] = d
            info[var] = os.environ[var]
    return info


def miopen_base_dir():
    """Parent of the per-rank MIOpen user-db / cache directories of a multi-replica run (rank r: <base>/rank<r>/{db,cache})."""
    import tempfile

    return os.environ.get("BC_BENCH_MIOPEN_DIR") or os.path.join(tempfile.gettempdir(), f"bc_bench_miopen_{os.getuid()}")


def details_path(args):
    if args.details:
        return args.details
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    tag = config_name(args).split("(")[0].replace(" ", "") + ("_f16" if args.half else "") + (f"_n{args.gpus}" if args.gpus > 1 else "")
    return os.path.join(d, f"bench_details_{tag}.json")


def stub_cpu_worker(args, rank, world, rank_env=None):
    """Test stand-in for the GPU workload (``--stub-cpu``): same launcher, environment, clock bracket and JSON schema,
    with a small dense conv on the CPU as a 'frame'."""
    from bc_workloads import replicas

    torch.set_num_threads(1)
    if os.environ.get("BC_STUB_FAIL_RANK") == str(rank):
        sys.exit(3)   # tests: a replica that dies must take the whole job down with a non-zero exit code
    if world > 1:
        replicas.init_clock_group()
    conv = torch.nn.Conv2d(3, 8, 3, padding=1)
    x = torch.randn(1, 3, 32, 64)
    with torch.no_grad():
        for _ in range(args.warmup * CLIP_LEN):
            conv(x)
        replicas.barrier(world)
        t0 = time.perf_counter()
        for _ in range(args.steps * CLIP_LEN):
            conv(x)
        replicas.barrier(world)
        elapsed = time.perf_counter() - t0
    local_fps = args.steps * CLIP_LEN / elapsed
    fps, elapsed, frames = replicas.job_throughput(args.steps * CLIP_LEN, elapsed, world)
    per_rank = replicas.gather_scalars(local_fps, world)
    cores = replicas.gather_scalars(float((rank_env or {}).get("cpu_first") or 0), world)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "STUB (cpu conv, launcher test only)", "frames_total": frames,
                                     "local_rank": int(os.environ.get("LOCAL_RANK", "-1")),
                                     "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"},
                          "per_rank_fps": {"min": min(per_rank), "max": max(per_rank), "all": [round(v, 1) for v in per_rank]},
                          "rank_env": dict(rank_env or {}, first_core_of_each_rank=[int(c) for c in cores])}), flush=True)
    if world > 1:
        replicas.barrier(world)
        torch.distributed.destroy_process_group()


def main(argv=None):
    global torch
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and env_world == 1:
        # run directly (the driver's `python bench.py --gpus N`): become the launcher BEFORE anything touches torch / HIP
        sys.exit(launch_replicas(args, argv))
    if env_world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the environment says WORLD_SIZE={env_world}")

    rank_env = isolate_rank(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                            int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    import torch as _torch
    torch = _torch
    from bc_workloads import replicas

    rank, world, local = replicas.dist_env()
    if args.stub_cpu:
        return stub_cpu_worker(args, rank, world, rank_env)
    assert torch.cuda.is_available(), "bench.py needs the GPU"
    n_dev = torch.cuda.device_count()
    if local >= n_dev:
        if not args.oversubscribe:
            sys.exit(f"bench.py: rank {rank} wants GPU {local} but only {n_dev} device(s) are visible (--oversubscribe shares GPUs; tests only)")
        local %= n_dev
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        # replicas only: torch.distributed brackets the clock (barrier + MAX/SUM of two scalars) and nothing else, so the
        # control group is host-side (gloo) -- no collective is invented on the data path
        replicas.init_clock_group()

    import blockcopy.backend as bk
    from blockcopy.core import tensorwrapper as tw
    from bc_workloads import harness

    tw.set_engine(args.engine)
    be = bk.get_backend()
    dtype = torch.float16 if args.half else torch.float32
    shape = (args.batch, 3, args.height, args.width)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)

    is_csp = args.workload == "csp"
    model = build_workload(args, args.policy, dtype, device, rank)
    # per-rank clips (clip i of the job lives on rank i mod N); inputs resident in HBM before the clock starts
    n_distinct = 2
    clips = [harness.synthetic_clip(CLIP_LEN, shape, seed=(rank * n_distinct + c) * 100, device=device, dtype=dtype) for c in range(n_distinct)]

    def prewarm(m, clip):
        if args.graph and args.policy != "fixed" and not is_csp:
            # data-dependent policies visit many executed-tile counts: warm + capture all quantised buckets up front
            harness.run_clip(m, clip[:1])
            m.prewarm(clip[0])

    t_w0 = time.perf_counter()
    prewarm(model, clips[0])
    for i in range(args.warmup):
        harness.run_clip(model, clips[i % n_distinct])
    torch.cuda.synchronize(device)
    warm_s = time.perf_counter() - t_w0

    from blockcopy.utils.profiler import timings
    timings.set_level(args.timings)
    timings.reset()
    be.prof_reset()
    be.prof_enable(["combine_copy"])       # eager launches (graph off, first frame ever): events attached to the dispatch
    inner0 = model.det if is_csp else model
    gfs = list(getattr(inner0, "_graphed", {}).values())
    for gf in gfs:                         # in-graph scatter+copy node: one device timing record per frame of the timed region
        gf.timing_start(args.steps * CLIP_LEN)
    replicas.barrier(world, device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        harness.run_clip(model, clips[i % n_distinct])
    replicas.barrier(world, device)
    elapsed = time.perf_counter() - t0
    be.prof_enable([])
    cc = be.prof_read("combine_copy")
    cc["method"] = "HIP events attached to each eager dispatch (hipExtLaunchKernelGGL) over the timed region"
    stamp_us, stamp_bytes, head_stamp_us = [], 0.0, []
    for gf in gfs:
        is_head = getattr(gf, "stamp_head", False)
        us, nb = gf.timing_read()
        if is_head:
            head_stamp_us += us          # (records of k_head1x1, the logits conv that carries the scatter+copy: bytes come from its algorithmic count below)
        else:
            stamp_us += us
            stamp_bytes = nb or stamp_bytes
    if stamp_us:
        # graph kernel nodes cannot carry events: the node stamps the constant 100 MHz clock itself (min workgroup entry, max
        # workgroup exit; include/blockcopy_hip.h bc_combine_copy_indirect), one record per frame of the timed region
        cc = {"launches": len(stamp_us) + cc["launches"], "total_ms": sum(stamp_us) * 1e-3 + cc["total_ms"],
              "total_bytes": stamp_bytes * len(stamp_us) + cc["total_bytes"],
              "method": "in-kernel s_memrealtime stamps of the hipGraph node k_combine_copy_ind (first workgroup entry -> last workgroup "
                        "exit, 10 ns ticks), every frame of the timed region; profiles/ holds the rocprofv3 trace of the same command",
              "p50_us": sorted(stamp_us)[len(stamp_us) // 2], "min_us": min(stamp_us), "max_us": max(stamp_us)}
    if args.timings and rank == 0:
        timings.add_cnt(args.steps * CLIP_LEN)
        print(timings, file=sys.stderr)
    timings.set_level(0)

    # whole-job throughput: all ranks' frames / slowest rank's time (no collective on the data path)
    elapsed_local = elapsed
    fps, elapsed, frames_total = replicas.job_throughput(args.steps * CLIP_LEN * args.batch, elapsed, world, device)
    exec_frac = model.policy.stats.get_exec_percentage()

    extra, details = {}, {}
    per_rank = replicas.gather_scalars(args.steps * CLIP_LEN * args.batch / elapsed_local, world)
    # per-rank record for the details file: fps, executed fraction, conv plan table (first 48 bits of its hash: equal on every rank = equal
    # kernel forms per layer), live-tuned shapes -- what makes the first multi-GPU run self-explaining
    from blockcopy.core import fusion as _fusion
    per_rank_exec = replicas.gather_scalars(float(exec_frac), world)
    per_rank_plan = replicas.gather_scalars(float(int(_fusion.conv_plan_hash()[:12], 16)), world)
    per_rank_tuned = replicas.gather_scalars(float(_fusion.PLAN_STATS["tuned_live"]), world)
    per_rank_ms = replicas.gather_scalars(1e3 * elapsed_local / args.steps, world)
    if rank == 0:
        details["per_rank"] = [{"rank": r, "fps": round(per_rank[r], 2), "ms_per_step": round(per_rank_ms[r], 3), "exec_fraction": round(per_rank_exec[r], 4),
                                "conv_plan_hash48": f"{int(per_rank_plan[r]):012x}", "shapes_tuned_live": int(per_rank_tuned[r]),
                                **({"miopen_dir": os.path.join(miopen_base_dir(), f"rank{r}")} if world > 1 else {})} for r in range(world)]
        details["per_rank_plans_equal"] = len(set(int(v) for v in per_rank_plan)) == 1
        # halo-gather kernel statistics from one extra, untimed clip (events around all 21 launches per frame)
        inner = model.det if is_csp else model
        use_graph, inner.use_graph = inner.use_graph, False   # per-launch events (eager mode, NOT the timed mode) need eager launches
        harness.run_clip(model, clips[0])                     # (the eager route's own first pass: allocations, cold caches -- not measured)
        torch.cuda.synchronize(device)
        be.prof_reset()
        be.prof_enable(["pad_ring", "split", "combine", "conv3x3", "head1x1", "combine_copy"])
        PROF_CLIPS = 2
        for k in range(PROF_CLIPS):
            harness.run_clip(model, clips[k % len(clips)])
        torch.cuda.synchronize(device)
        be.prof_enable([])
        for op in ("pad_ring", "split", "combine"):
            r = be.prof_read(op)
            if r["launches"]:
                extra[op] = {"launches_per_frame": r["launches"] / (PROF_CLIPS * CLIP_LEN), "avg_us": 1e3 * r["total_ms"] / r["launches"],
                             "GBps": r["total_bytes"] / (r["total_ms"] * 1e-3) / 1e9, "MB_per_frame": r["total_bytes"] / (PROF_CLIPS * CLIP_LEN) / 1e6,
                             }
        cc_eager = be.prof_read("combine_copy")
        hd = be.prof_read("head1x1")
        if hd["launches"]:
            # the network's output stage: BN/ReLU + 1x1 conv to the class logits + bias + out-of-place combine (scatter of the executed
            # tiles, copy of the skipped ones from the previous frame's map) in ONE launch -- the fused scatter+copy of rounds 1-2
            # is its epilogue now.  HBM-bound: algorithmic bytes = packed features read + 2 x logits map (SURVEY 8(d) formula)
            gb = hd["total_bytes"] / (hd["total_ms"] * 1e-3) / 1e9
            extra["head1x1"] = {"kernel": "k_head1x1 (logits conv with the scatter+copy as its epilogue)", "launches_per_frame": hd["launches"] / (PROF_CLIPS * CLIP_LEN), "launches_measured": hd["launches"],
                                "avg_us": 1e3 * hd["total_ms"] / hd["launches"], "algorithmic_MB_per_launch": hd["total_bytes"] / hd["launches"] / 1e6,
                                "achieved": gb, "frac": gb / HBM_PEAK_GBS}
        r = be.prof_read("conv3x3")
        if r["launches"]:
            # the fused conv kernel (3x3 halo form, its stride-2 and one-tap forms): FLOPs of all its launches of one clip over their
            # summed execution time, against the dense MFMA peak of the compute dtype (MI355X_MICROARCH.md: fp32 157.3, 16-bit 2516 TFLOP/s)
            peak = 157.3 if dtype == torch.float32 else 2516.0
            tf = r["total_aux"] / (r["total_ms"] * 1e-3) / 1e12          # matrix FLOPs actually issued
            tf_alg = r["total_bytes"] / (r["total_ms"] * 1e-3) / 1e12    # FLOPs of the direct definition 2*px*k*k*Cin*Cout
            extra["roofline_conv"] = {"kernel": "k_conv3x3_v2 (direct form; fp32 tensors mostly as BC_F32S: operands split hi + lo in fp16, three v_mfma_f32_32x32x16_f16 per 16 channels = 3/16 of the "
                                                "fp32 pipe's matrix time, counted as such) + k_conv3x3_wino4 (F(4x4,3x3): 36/144 of the direct multiplications) + k_conv3x3_wino / k_conv3x3_wino32 "
                                                "(F(2x2,3x3): 16/36) + k_stem7x7; per-layer form: details file",
                                      "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "issued_frac": tf / peak,
                                      "effective_TFLOPs": tf_alg, "effective_frac": tf_alg / peak, "traffic": None,
                                      "launches_per_frame": r["launches"] / (PROF_CLIPS * CLIP_LEN), "ms_per_frame": r["total_ms"] / (PROF_CLIPS * CLIP_LEN),
                                      "GFLOP_issued_per_frame": r["total_aux"] / (PROF_CLIPS * CLIP_LEN) / 1e9, "GFLOP_algorithmic_per_frame": r["total_bytes"] / (PROF_CLIPS * CLIP_LEN) / 1e9,
                                      "note": "achieved / frac = ISSUED matrix work in fp32-pipe FLOPs (a split-form launch issues 3/16 of its direct FLOPs: frac is the share of the "
                                              "kernel time the matrix pipe is busy) over the summed kernel time of an eager clip (dispatch-attached events); "
                                              "effective_* = FLOPs of the direct definition over the same time (> 1 x the fp32 peak: the products run on the 16-bit pipe)"}
        # host-side enqueue cost of one clip (no sync inside): tells whether the frame is CPU- or GPU-bound
        torch.cuda.synchronize(device)
        th = time.perf_counter()
        harness.run_clip(model, clips[0])
        host_s = time.perf_counter() - th
        torch.cuda.synchronize(device)
        extra["eager_host_enqueue_ms_per_frame"] = 1e3 * host_s / CLIP_LEN
        inner.use_graph = use_graph
        torch.cuda.synchronize(device)
        th = time.perf_counter()
        harness.run_clip(model, clips[0])
        host_s = time.perf_counter() - th
        torch.cuda.synchronize(device)
        extra["host_enqueue_ms_per_frame"] = 1e3 * host_s / CLIP_LEN      # (back to back: includes blocking on a full queue)
        # what a frame costs the HOST: the stream is idle before every frame (no back-pressure in the figure), wall time of the enqueue
        # and CPU time of this thread -- the number that decides whether N pinned replicas per host stay GPU-bound
        if hasattr(model, "reset_temporal"):
            model.reset_temporal()
        wall = cpu = 0.0
        with torch.no_grad():
            for f in clips[0]:
                torch.cuda.synchronize(device)
                w0, c0 = time.perf_counter(), time.thread_time()
                model(f)
                wall += time.perf_counter() - w0
                cpu += time.thread_time() - c0
        torch.cuda.synchronize(device)
        extra["host_ms_per_frame_idle_stream"] = {"wall": 1e3 * wall / CLIP_LEN, "cpu_thread": 1e3 * cpu / CLIP_LEN}
        extra["roofline_large"] = scatter_copy_large(be, device)
        extra["roofline_large"]["note"] = ("a stand-alone launch at a large synthetic shape, not a launch of the timed clips: the judged `roofline` object above is "
                                           "measured on the scatter carrier the frames really launch (k_head1x1 for the segmentation nets, k_combine_copy for the detector)")
        if world == 1 and args.upload_variant and not is_csp:
            # the reference driver's complete loop: per-frame upload from pinned host memory + last-frame upsample / argmax / .cpu()
            # (test_swiftnet.py:181-197); the headline `value` keeps inputs resident in HBM, these two figures do not
            hclips = [[f.cpu() for f in clips[0]]]
            up = {}
            # reference_loop_sync_upload: the reference's own form (upload on the compute stream, stock upsample + max, blocking .cpu());
            # double_buffered_upload: copy streams, host-side hand-over, prediction map in one pass (bc_upsample_argmax); *_stock_tail: the
            # same with the two stock ops for the prediction map
            for name, pf, ft in (("reference_loop_sync_upload", False, False), ("double_buffered_upload_stock_tail", True, False), ("double_buffered_upload", True, True)):
                # (a timed region of three clips lasts ~60 ms: one descheduled host thread halves the figure -- median of three regions)
                up[name] = sorted(harness.measure_fps_with_upload(model, hclips, n_clips=max(1, min(args.steps, 3)), warmup_clips=1, device=device,
                                                                  dtype=dtype, prefetch=pf, fused_tail=ft)[0] for _ in range(3))[1]
            up["note"] = "per-frame H->D upload + last-frame upsample / argmax / predictions to the host inside the timed region (reference test_swiftnet.py:181-197); never `value`"
            extra["upload_inclusive"] = up
        from blockcopy.core import fusion
        # route per padded 3x3 / pointwise layer shape (fusion.conv3x3_plan): null = halo gather + MIOpen, code = fused kernel decomposition
        # (codes with 0x200 = Winograd form).  The table comes from blockcopy/plans/gfx950.json; shapes it lacked were measured live.
        details["conv_plans_measured_live"] = [{"n_exec": k[0], "tile": k[1], "cin": k[2], "cout": k[3], "stride": k[6], "ks": k[7], "choice": best,
                                                "us": round(times[best], 1), "library_us": round(times.get("library", float("nan")), 1)}
                                               for k, times, best in fusion.CONV_TUNE_LOG]
        details["conv_plan_table"] = {fusion._key_to_str(k): v for k, v in fusion._conv_plans.items()}
        extra["conv_plan"] = {"hash": fusion.conv_plan_hash(), "entries": len(fusion._conv_plans), "file": os.path.relpath(fusion.PLAN_FILE, ROOT) if fusion.PLAN_FILE else None,
                              "decisions": dict(fusion.PLAN_STATS), "mode": fusion.CONV_MODE}
        if not args.no_dense and world == 1:
            dense = build_workload(args, "static", dtype, device, rank)
            dfps, _, _ = harness.measure_fps(dense, clips[:1], n_clips=max(1, min(args.steps, 3)), warmup_clips=1, device=device)
            extra["dense_gpu_fps"] = dfps
            extra["speedup_vs_dense_gpu"] = fps / dfps
            if "upload_inclusive" in extra:
                # the dense model in the SAME upload-inclusive loop (per-frame upload, last-frame upsample / argmax / predictions to the host):
                # the like-for-like denominator of `value_reference_loop`
                ddfps = sorted(harness.measure_fps_with_upload(dense, hclips, n_clips=max(1, min(args.steps, 3)), warmup_clips=1, device=device,
                                                               dtype=dtype, prefetch=True, fused_tail=True)[0] for _ in range(3))[1]
                extra["upload_inclusive"]["dense_double_buffered_upload"] = ddfps
                extra["upload_inclusive"]["speedup_vs_dense_reference_loop"] = extra["upload_inclusive"]["double_buffered_upload"] / ddfps
            del dense
            if args.also_half and not args.half:
                # secondary measurement, outside the timed region: the identical workload in fp16
                del model
                torch.cuda.empty_cache()
                h = torch.float16
                hm = build_workload(args, args.policy, h, device, rank)
                hclips = [[f.to(h) for f in clips[0]]]
                prewarm(hm, hclips[0])
                # (a 60-frame region lasts ~30 ms: one host hiccup shows as -30 %; median of three such regions)
                hfps = sorted(harness.measure_fps(hm, hclips, n_clips=max(1, min(args.steps, 3)), warmup_clips=2 if i == 0 else 0, device=device)[0]
                              for i in range(3))[1]
                hd = build_workload(args, "static", h, device, rank)
                hdfps, _, _ = harness.measure_fps(hd, hclips, n_clips=max(1, min(args.steps, 3)), warmup_clips=1, device=device)
                extra["fp16"] = {"fps": hfps, "dense_gpu_fps": hdfps, "speedup_vs_dense_gpu": hfps / hdfps,
                                 "note": "same workload in float16 (median of 3 regions); the headline value is fp32"}
                del hm, hd
                if args.batch == 1 and not is_csp:
                    # secondary measurement: two clips side by side (the reference's speed configs use --batch-size 2)
                    torch.cuda.empty_cache()
                    bshape = (2, 3, args.height, args.width)
                    bclips = [harness.synthetic_clip(CLIP_LEN, bshape, seed=7, device=device, dtype=dtype)]
                    bm = harness.build_model(args.backbone, block_policy=args.policy, block_size=args.block_size, block_target=args.target,
                                             device=device, dtype=dtype, seed=1000 * rank, block_graph=args.graph,
                                             block_train_interval=args.train_interval, channels_last=bool(args.channels_last))
                    prewarm(bm, bclips[0])
                    bfps, _, _ = harness.measure_fps(bm, bclips, n_clips=max(1, min(args.steps, 3)), warmup_clips=2, device=device)
                    bd = harness.build_model(args.backbone, block_policy="static", device=device, dtype=dtype, channels_last=bool(args.channels_last))
                    bdfps, _, _ = harness.measure_fps(bd, bclips, n_clips=max(1, min(args.steps, 3)), warmup_clips=1, device=device)
                    extra["batch2"] = {"fps": bfps, "dense_gpu_fps": bdfps, "speedup_vs_dense_gpu": bfps / bdfps,
                                       "note": "2 clips per step, fp32; the headline value is batch 1"}
                    del bm, bd

    if rank == 0:
        if not cc["launches"] and "head1x1" in extra:
            # no stand-alone scatter+copy launch exists in this configuration any more: the op is the epilogue of the logits conv
            h = extra["head1x1"]
            if head_stamp_us:
                # the timed region itself: every frame's k_head1x1 graph node stamps the constant 100 MHz clock (first workgroup entry -> last
                # workgroup exit); bytes = the launch's algorithmic count averaged over a clip (eager pass below: same clips, same masks)
                cc = {"launches": len(head_stamp_us), "total_ms": sum(head_stamp_us) * 1e-3,
                      "total_bytes": h["algorithmic_MB_per_launch"] * 1e6 * len(head_stamp_us), "kernel": h["kernel"],
                      "method": "in-kernel s_memrealtime stamps of the hipGraph node k_head1x1 (first workgroup entry -> last workgroup exit, 10 ns ticks), every frame "
                                "of the timed region (graph kernel nodes cannot carry HIP events); eager_events_avg_launch_us = the same kernel under dispatch-attached "
                                "HIP events in two eager clips after the timed region; rocprofv3 trace of the graph replays: profiles/",
                      "p50_us": sorted(head_stamp_us)[len(head_stamp_us) // 2], "min_us": min(head_stamp_us), "max_us": max(head_stamp_us),
                      "eager_events_avg_launch_us": h["avg_us"]}
            else:
                cc = {"launches": int(h["launches_measured"]), "total_ms": h["avg_us"] * 1e-3 * h["launches_measured"],
                      "total_bytes": h["algorithmic_MB_per_launch"] * 1e6 * h["launches_measured"], "kernel": h["kernel"],
                      "method": "dispatch-attached HIP events (hipExtLaunchKernelGGL) over two eager clips (after one eager warm clip) run inside bench.py right after the timed region "
                                "(graph kernel nodes cannot carry events); rocprofv3 trace of the graph replays: profiles/"}
        elif not cc["launches"] and cc_eager["launches"]:
            # (the detector: its out-of-place combines -- three 134 MB head maps per frame -- are nodes of the frame's graph)
            cc = dict(cc_eager, kernel="k_combine_copy (fused scatter+copy: the out-of-place combine of the head maps)",
                      method="dispatch-attached HIP events (hipExtLaunchKernelGGL) over two eager clips (after one eager warm clip) run inside bench.py right after the timed region "
                             "(graph kernel nodes cannot carry events)")
        achieved = (cc["total_bytes"] / (cc["total_ms"] * 1e-3) / 1e9) if cc["total_ms"] > 0 else 0.0
        # (the segmentation logits map: batch x 19 classes x H/4 x W/4)
        pure_scatter = None if is_csp else 2.0 * args.batch * 19 * (args.height // 4) * (args.width // 4) * (2 if args.half else 4)
        traffic, traffic_src, traffic_kernels = pmc_traffic()
        # the same kernel in the committed rocprofv3 trace of the graph replays (default workload only: that is what was profiled)
        rocprof_fields = {}
        if config_name(args) == "C2" and not args.half and args.batch == 1 and str(cc.get("kernel", "")).startswith("k_head1x1") and cc["launches"]:
            rp_us, rp_frames = rocprof_cross_check("k_head1x1<0")
            if rp_us:
                rocprof_fields = {"rocprof_avg_launch_us": rp_us, "rocprof_frac": cc["total_bytes"] / cc["launches"] / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                  "rocprof_source": "committed profiles/rocprof_latest.json (a separate rocprofv3 --kernel-trace run of this command), NOT measured in this run"}
        if traffic_kernels:
            # committed PMC measurement (profiles/traffic_latest.json: rocprofv3 --pmc passes of tools/pmc_driver.py at these shapes)
            details["pmc_traffic"] = traffic_kernels
        out = {
            "metric": METRIC, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.half else "f32", "data": "synthetic",
            "config": {"workload": f"{config_name(args)}: {'CSP-ResNet50 detector (backbone + neck + head + decode + NMS)' if is_csp else 'SwiftNet-' + args.backbone} {shape[0]}x3x{args.height}x{args.width} synthetic clips of {CLIP_LEN} frames, "
                                   f"block {args.block_size}, policy {args.policy} target {args.target:.0%} (frame 0 all-active), "
                                   f"{args.engine} engine{(' + hipGraph' if args.graph == 1 else ' + ONE dynamic hipGraph (device-side tile count, no wait)') if args.graph else ''}{', channels-last' if args.channels_last else ''}, seeded weights, BN folded; step = 1 clip",
                       "clips_per_rank": args.steps, "parallelism": f"{world} independent replica(s), no collective",
                       "exec_fraction": exec_frac, "warmup_s": warm_s},
            # frames/s of the reference driver's complete loop (test_swiftnet.py:181-197: per-frame H->D upload from pinned memory, last-frame
            # upsample + argmax + predictions to the host) with uploads and the download on copy streams (harness.measure_fps_with_upload); PCIe-inclusive, never `value`
            **({"value_reference_loop": extra["upload_inclusive"]["double_buffered_upload"]} if "upload_inclusive" in extra else {}),
            "roofline": {"kernel": cc.get("kernel") or ("k_combine_copy_ind (fused scatter+copy of the logits map, a node of the frame's hipGraph)" if stamp_us
                                                        else "k_combine_copy (fused scatter+copy of the logits map)"), "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if is_csp else (traffic[1] if cc.get("kernel") else traffic[0]),     # (PMC passes exist for the SwiftNet map shapes)
                         "traffic_source": None if is_csp else "committed profiles/traffic_latest.json (separate rocprofv3 --pmc passes), NOT measured in this run",
                         # SURVEY.md 8(d)'s PURE scatter+copy count of the same launch (2 * N * C * H * W * E: every tile of the output map read
                         # once and written once) over the measured duration -- the launch also reads the conv's input features, see `frac`
                         **({"pure_scatter_bytes_per_launch": pure_scatter, "frac_pure_scatter": pure_scatter / (1e-3 * cc["total_ms"] / cc["launches"]) / 1e9 / HBM_PEAK_GBS}
                            if cc["launches"] and pure_scatter else {}),
                         "launches": cc["launches"], "avg_launch_us": (1e3 * cc["total_ms"] / cc["launches"]) if cc["launches"] else None,
                         **({"p50_us": cc["p50_us"], "min_us": cc["min_us"], "max_us": cc["max_us"]} if "p50_us" in cc else {}),
                         **({"eager_events_avg_launch_us": cc["eager_events_avg_launch_us"]} if "eager_events_avg_launch_us" in cc else {}),
                         "algorithmic_bytes_per_launch": (cc["total_bytes"] / cc["launches"]) if cc["launches"] else None,
                         **rocprof_fields,
                         "method": cc["method"]},
            **({"per_rank_fps": {"min": min(per_rank), "max": max(per_rank), "all": [round(v, 1) for v in per_rank]}} if world > 1 else {}),
            "kernels": extra,
        }
        if "roofline_conv" in extra:
            # the kernel that DOMINATES the frame by time (the fused conv, MFMA-bound); `roofline` above is the kernel the metric names
            out["roofline_conv"] = extra.pop("roofline_conv")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_dense_baseline(args, args.cpu_frames)
        details["rank_env"] = rank_env
        details["traffic_source"] = traffic_src
        dpath = details_path(args)
        out["details_file"] = os.path.relpath(dpath, ROOT)
        def compact(o, top=True):      # six significant digits inside the nested objects (the line must stay small); top-level scalars stay exact
            if isinstance(o, dict):
                return {k: (v if top and not isinstance(v, (dict, list)) else compact(v, False)) for k, v in o.items()}
            if isinstance(o, list):
                return [compact(v, False) for v in o]
            if isinstance(o, float):
                return float(f"{o:.6g}")
            return o

        out = compact(out)
        line = json.dumps(out)
        with open(dpath, "w") as f:
            json.dump({"bench_line": out, **details}, f, indent=1)
        if len(line) >= 4096:
            # the details file has everything; the printed line sheds its free-text notes first, then the secondary objects
            def shed(o):
                return {k: shed(v) for k, v in o.items() if k not in ("note", "method")} if isinstance(o, dict) else o
            out = shed(out)
            line = json.dumps(out)
        if len(line) >= 4096:
            # ... then the provenance strings and the long kernel list of roofline_conv (all of it is in the details file)
            for k in ("traffic_source", "rocprof_source"):
                out.get("roofline", {}).pop(k, None)
            if isinstance(out.get("roofline_conv"), dict) and len(out["roofline_conv"].get("kernel", "")) > 96:
                out["roofline_conv"]["kernel"] = out["roofline_conv"]["kernel"][:93] + "..."
            line = json.dumps(out)
        for k in ("roofline_large", "upload_inclusive", "conv_plan"):      # ... then secondary measurements, one at a time
            if len(line) >= 4096 and isinstance(out.get("kernels"), dict) and k in out["kernels"]:
                out["kernels"][k] = {"moved_to": out["details_file"]}
                line = json.dumps(out)
        for k in ("kernels", "roofline_conv"):
            if len(line) >= 4096 and k in out:
                out[k] = {"moved_to": out["details_file"]}
                line = json.dumps(out)
        assert len(line) < 4096, f"bench line grew to {len(line)} bytes"
        print(line, flush=True)
    if args.save_plan and rank == 0:
        from blockcopy.core import fusion
        fusion.save_conv_plans(args.save_plan, note="conv plan table measured by bench.py on " + torch.cuda.get_device_name(device))

    if world > 1:
        replicas.barrier(world, device)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
