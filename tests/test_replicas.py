"""Clip-parallel replicas (the N>1 path): world_size-2 gloo processes on CPU.  Each rank runs its share of the clips
through the block engine (oracle-backed on CPU); results must equal a single-process run of the same clips and the
job throughput must be SUM(frames) / MAX(time) with no collective on the data path."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_clip(clip_id):
    import blockcopy
    from blockcopy.core.argparser import default_settings
    from bc_workloads import seeded
    from test_host_logic import TinyNet

    net = TinyNet()
    net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()))
    model = blockcopy.BlockCopyModel(net.eval(), default_settings(block_policy="fixed", block_size=8, block_target=0.5, block_seed=clip_id))
    model.reset_temporal()
    outs = []
    with torch.no_grad():
        for t in range(4):
            outs.append(model(seeded.synthetic_frame(1000 * clip_id + t, (1, 3, 32, 48))).clone())
    return torch.stack(outs)


def _worker(rank, world, port, n_clips, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(root, "blockcopy-video-processing-pytorch_amd"), os.path.join(root, "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import blockcopy.backend as bk
    from oracle_backend import OracleBackend
    from bc_workloads import replicas

    bk.set_backend(OracleBackend())
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    assert replicas.dist_env() == (rank, world, rank)
    mine = replicas.clips_for_rank(n_clips, rank, world)
    replicas.barrier(world)
    outs = {i: _run_clip(i) for i in mine}
    # fake but rank-dependent timings: rank r "took" (r + 1) seconds for its 4-frame clips
    fps, t_max, frames = replicas.job_throughput(4 * len(mine), float(rank + 1), world)
    replicas.barrier(world)
    ret[rank] = (mine, {i: o.numpy() for i, o in outs.items()}, fps, t_max, frames)
    torch.distributed.destroy_process_group()


def test_two_replicas_match_single_process(oracle_backend):
    from bc_workloads import replicas

    world, n_clips = 2, 5
    assert replicas.clips_for_rank(n_clips, 0, 2) == [0, 2, 4] and replicas.clips_for_rank(n_clips, 1, 2) == [1, 3]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    seen = {}
    for r in range(world):
        mine, outs, fps, t_max, frames = ret[r]
        assert frames == 4 * n_clips and t_max == 2.0 and fps == pytest.approx(4 * n_clips / 2.0)
        seen.update(outs)
    assert sorted(seen) == list(range(n_clips))
    for i in range(n_clips):
        assert torch.equal(torch.from_numpy(seen[i]), _run_clip(i)), f"clip {i} depends on its placement"


def test_single_process_throughput():
    from bc_workloads import replicas

    assert replicas.job_throughput(40, 2.0, 1) == (20.0, 2.0, 40)
