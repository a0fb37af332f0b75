"""Randomised-geometry parity sweep shared by tests/test_gpu_ops.py: random (N, C, GH, GW, bs, pad, dtype, layout) and
random execution grids over a multi-frame chain, HIP library vs the CPU oracle, bit-exact.  Also runnable as a script
(`python tests/random_geometry.py SEED COUNT`) so that the test can repeat it in child processes with
BC_HALO_KERNEL=rows|lds|simple (the library reads that variable once per process)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for _p in (os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"), os.path.join(ROOT, "oracle"), HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)

DTYPES = [torch.float32, torch.float16, torch.bfloat16]


def random_case(rng):
    bs = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 20, 32, 48, 64]))
    pad = int(rng.integers(1, min(bs, 3) + 1))
    N = int(rng.choice([1, 1, 2, 3]))
    GH, GW = int(rng.integers(1, 6)), int(rng.integers(1, 7))
    C = int(rng.choice([1, 2, 3, 5, 8, 16, 19, 24, 33, 64]))
    while N * GH * GW * C * (bs + 2 * pad) ** 2 > 3_000_000:   # keep the CPU oracle in the millisecond range
        C = max(1, C // 2)
        if C == 1:
            GH, GW = max(1, GH - 1), max(1, GW - 1)
    return N, C, GH, GW, bs, pad


def random_grid(rng, total, t):
    if t == 0:
        return np.ones(total, bool)
    frac = rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])
    g = rng.random(total) < frac
    if not g.any():
        g[rng.integers(total)] = True
    return g


def sweep(seed, count, frames=4):
    import blockcopy.backend as bk
    import oracle as O
    from oracle_backend import OracleBackend

    be, chk = bk.get_backend(), OracleBackend()
    assert be.name == "hip"
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    done = []
    for _ in range(count):
        N, C, GH, GW, bs, p = case = random_case(rng)
        dtype = DTYPES[int(rng.integers(len(DTYPES)))]
        nhwc = bool(rng.integers(2)) and (C * torch.empty((), dtype=dtype).element_size()) % 2 == 0
        lay = (lambda x: x.contiguous(memory_format=torch.channels_last)) if nhwc else (lambda x: x.contiguous())
        H, W, T = GH * bs, GW * bs, N * GH * GW
        ring_dev = torch.zeros((T, C, 4 * p * bs), dtype=dtype).cuda()
        ring_cpu = torch.zeros((T, C, 4 * p * bs), dtype=dtype)
        tag = (case, str(dtype), "nhwc" if nhwc else "nchw")
        for t in range(frames):
            grid = random_grid(rng, T, t).reshape(N, 1, GH, GW)
            gi, m = O.c_grid_mappings(grid)
            gi_d, m_d = torch.from_numpy(gi).cuda(), torch.from_numpy(m).cuda()
            image = torch.randn((N, C, H, W), generator=gen).to(dtype)
            prev = torch.randn((N, C, H, W), generator=gen).to(dtype)
            want = torch.empty((len(m), C, bs, bs), dtype=dtype)
            O.c_split(want, image, m)
            got = lay(torch.zeros((len(m), C, bs, bs), dtype=dtype).cuda())
            be.split(got, lay(image.cuda()), m_d, gi_d)
            assert torch.equal(got.cpu().contiguous(), want), ("split", tag, t)
            want_out = prev.clone()
            O.c_combine(want, want_out, m)
            fused = lay(torch.zeros((N, C, H, W), dtype=dtype).cuda())
            be.combine_copy(got, lay(prev.cuda()), fused, gi_d)
            assert torch.equal(fused.cpu().contiguous(), want_out), ("combine_copy", tag, t)
            inplace = lay(prev.cuda())
            be.combine(got, inplace, gi_d, m_d)
            assert torch.equal(inplace.cpu().contiguous(), want_out), ("combine", tag, t)
            feats = torch.randn((len(m), C, bs, bs), generator=gen).to(dtype)
            wantp = chk.pad_ring(feats, ring_cpu, torch.from_numpy(gi), torch.from_numpy(m), p)
            gotp = be.pad_ring(lay(feats.cuda()), ring_dev, gi_d, m_d, p)
            assert torch.equal(gotp.cpu().contiguous(), wantp), ("pad_ring", tag, t)
        done.append(tag)
    torch.cuda.synchronize()
    return done


if __name__ == "__main__":
    cases = sweep(int(sys.argv[1]), int(sys.argv[2]))
    print(f"random_geometry ok: {len(cases)} cases, BC_HALO_KERNEL={os.environ.get('BC_HALO_KERNEL', 'auto')}")
