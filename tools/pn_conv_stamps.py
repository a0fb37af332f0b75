"""In-kernel stamps of the policy net's conv kernel on one layer shape: where a workgroup's life goes (100 MHz s_memrealtime).
python tools/pn_conv_stamps.py [--hw 256 512] [--cin 32 --cout 32 --stride 1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, nargs=2, default=[256, 512])
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--precision", type=int, default=1)
    args = ap.parse_args()
    import blockcopy.backend as bk

    lib = bk.get_backend().lib
    N, (H, W), Ci, Co, s = args.n, args.hw, args.cin, args.cout, args.stride
    Hy, Wy = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn((N, H, W, Ci), device="cuda")
    w = torch.randn((9, Ci, Co), device="cuda") * 0.1
    out = torch.empty((N, Hy, Wy, Co), device="cuda")
    n_part = lib.bc_pn_conv_partials(N, Hy, Wy, Co)
    stats = torch.zeros(n_part * 2 * Co, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    n_wg = n_part * (Co // 32)
    stamps = torch.zeros((n_wg, 8), dtype=torch.int64, device="cuda")

    def launch():
        assert lib.bc_pn_conv_nhwc(out.data_ptr(), x.data_ptr(), w.data_ptr(), N, H, W, Ci, Hy, Wy, Co, 3, s, 0, None, None, 0, None, None, 0, stats.data_ptr(),
                                   stats.numel(), args.precision, st) == 0

    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        launch()
    b.record()
    torch.cuda.synchronize()
    print(f"conv {Ci}->{Co} stride {s} on {N}x{H}x{W}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per launch, {n_wg} workgroups")
    lib.bc_pn_set_stamps(stamps.data_ptr())
    launch()
    torch.cuda.synchronize()
    lib.bc_pn_set_stamps(None)
    t = stamps.cpu().numpy().astype(np.float64)[:, :6] * 0.01      # us
    t0 = t[:, 0].min()
    names = ["entry", "first patch requested", "first stage in LDS", "first tile multiplied", "first tile stored", "all tiles done"]
    print(f"launch window by stamps: {t[:, 5].max() - t0:.1f} us;  workgroup entry spread: median {np.median(t[:, 0]) - t0:.1f}, max {t[:, 0].max() - t0:.1f} us")
    for k in range(1, 6):
        d = t[:, k] - t[:, k - 1]
        print(f"  {names[k - 1]:24s} -> {names[k]:24s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}")
    life = t[:, 5] - t[:, 0]
    print(f"  workgroup life: median {np.median(life):.2f} us, max {life.max():.2f} us")


if __name__ == "__main__":
    main()
