#!/usr/bin/env python3
"""Launches every hand-written kernel of the DEFAULT bench path (channels-last fp32, C2 shapes, 64 of 128 tiles executed)
plus the large scatter+copy shape a few times each, with nothing else on the GPU -- the command to put behind
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (one counter per pass; a PMC pass over the whole bench.py takes >15 min
because every library kernel is serialised).  Writes the manifest of what was launched, in launch order, with the
ALGORITHMIC bytes of each launch (SURVEY.md section 8(d) formulas) to gpurun_out/pmc_manifest.json; tools/pmc_traffic.py
joins it with the counter CSVs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from kbench import grid_tables  # noqa: E402

REPS = 5


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def halo_bytes(n, C, bs, p, E=4):
    """repad formula: read (bs+2p)^2 - z, write (bs+2p)^2 per tile and channel; z (image-border zeros) ignored here (upper bound),
    plus the ring refresh n*C*4*p*bs."""
    return n * C * (2 * (bs + 2 * p) ** 2 + 4 * p * bs) * E


def main():
    be = bk.get_backend()
    manifest = []

    def run(label, algorithmic_bytes, fn):
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        manifest.append({"label": label, "algorithmic_bytes": float(algorithmic_bytes), "launches": REPS})

    n = 64
    gi, m = grid_tables(1, 8, 16, n)
    # fused scatter+copy: C2 logits and the C5 head map
    for (C, H, W, bs, name) in [(19, 256, 512, 32, "combine_copy C2 logits (1,19,256,512)"), (256, 256, 512, 32, "combine_copy C5 head (1,256,256,512)")]:
        blocks = cl(torch.randn((n, C, bs, bs), device="cuda"))
        prev = cl(torch.randn((1, C, H, W), device="cuda"))
        out = torch.empty_like(prev)
        run(name, 2.0 * C * H * W * 4, lambda: be.combine_copy(blocks, prev, out, gi))
    # the network's output stage in one launch (round 3): BN/ReLU + 1x1 conv 128 -> 19 + bias + out-of-place combine of the C2 logits map.
    # algorithmic bytes = packed features read + skipped tiles read from the previous map + the whole map written + weights (the packed
    # logits of the executed tiles never exist in memory, so SURVEY 8(d)'s 2*N*C*H*W*E of the stand-alone scatter+copy shrinks by them)
    C, H, W, bs, cin = 19, 256, 512, 32, 128
    feats = cl(torch.randn((n, cin, bs, bs), device="cuda"))
    w = torch.randn((C, cin, 1, 1), device="cuda") * 0.1
    wpk = be.pack_head1x1_weights(w)
    sc, bias = torch.rand(cin, device="cuda") + 0.5, torch.randn(C, device="cuda")
    prev = cl(torch.randn((1, C, H, W), device="cuda"))
    out = torch.empty_like(prev)
    run("head1x1_scatter C2 logits (64,128,32,32) -> (1,19,256,512)", 4.0 * (n * cin * bs * bs + (128 - n) * C * bs * bs + C * H * W + 32 * cin),
        lambda: be.head1x1_scatter(feats, wpk, C, (sc, sc, True), bias, gi, m, prev=prev, out=out))
    blocks = cl(torch.randn((n, C, bs, bs), device="cuda"))
    slots = torch.tensor([prev.data_ptr(), out.data_ptr(), 0], dtype=torch.int64).cuda()
    run("combine_copy_indirect C2 logits (1,19,256,512)", 2.0 * C * H * W * 4, lambda: be.combine_copy_indirect(blocks, slots, gi, (1, C, H, W)))
    # gather / in-place scatter of the network input (NCHW frame)
    img = torch.randn((1, 3, 1024, 2048), device="cuda")
    blocks = torch.empty((n, 3, 128, 128), device="cuda")
    run("split network input (64,3,128,128)", 2.0 * n * 3 * 128 * 128 * 4, lambda: be.split(blocks, img, m, gi))
    run("combine_ frame_state (64,3,128,128)", 2.0 * n * 3 * 128 * 128 * 4, lambda: be.combine(blocks, img, gi, m))
    # the network-input stage of a graph-replayed frame (round 4): executed tiles of the caller's frame straight into the frame state,
    # source address through a device word
    src = torch.randn((1, 3, 1024, 2048), device="cuda")
    slot = torch.tensor([src.data_ptr()], dtype=torch.int64).cuda()
    run("tile_copy_indirect network input (64 tiles of 3x128x128)", 2.0 * n * 3 * 128 * 128 * 4, lambda: be.tile_copy_indirect(img, slot, m, 128))
    # channels-last halo gathers still on the default path: stride-2 conv inputs, stem pool
    for (C, bs, name) in [(64, 32, "pad_ring_nhwc layer2.0.conv1 input (64,64,32,32)"), (128, 16, "pad_ring_nhwc layer3.0.conv1 input (64,128,16,16)"),
                          (256, 8, "pad_ring_nhwc layer4.0.conv1 input (64,256,8,8)")]:
        feats = cl(torch.randn((n, C, bs, bs), device="cuda"))
        ring = torch.randn((128, C, 4 * bs), device="cuda")
        sc = torch.rand(C, device="cuda") + 0.5
        run(name, halo_bytes(n, C, bs, 1), lambda: be.pad_ring(feats, ring, gi, m, 1, (sc, sc, True)))
        add = cl(torch.randn((n, C, bs, bs), device="cuda"))
        run(name.replace("pad_ring_nhwc", "pad_ring_add_nhwc"), halo_bytes(n, C, bs, 1) + 2.0 * n * C * bs * bs * 4,
            lambda: be.pad_ring_add(feats, add, ring, gi, m, 1, (sc, sc, True)))
    feats = cl(torch.randn((n, 64, 64, 64), device="cuda"))
    ring = torch.randn((128, 64, 4 * 64), device="cuda")
    run("maxpool3x3s2_ring_nhwc stem (64,64,64,64)", n * 64 * 4 * ((64 + 1) ** 2 + 64 * 64 / 4 + 4 * 64),
        lambda: be.maxpool3x3s2_ring(feats, ring, gi, m, None))
    # fused halo+conv (CU-balanced kernel): bytes = input tiles + halo ring reads (~4*bs per tile and channel) + output + weights
    for (Cin, Cout, bs, name) in [(64, 64, 32, "conv3x3 layer1"), (128, 128, 16, "conv3x3 layer2"), (256, 256, 8, "conv3x3 layer3"),
                                  (512, 512, 4, "conv3x3 layer4"), (128, 128, 32, "conv3x3 up 1/4")]:
        feats = cl(torch.randn((n, Cin, bs, bs), device="cuda"))
        ring = torch.randn((128, Cin, 4 * bs), device="cuda")
        wpk = be.pack_conv3x3_weights(torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05)
        alg = 4.0 * (n * Cin * ((bs + 2) ** 2 + 4 * bs) + n * Cout * bs * bs + 9 * Cin * Cout)
        run(f"{name} ({n},{Cin}->{Cout},{bs}x{bs})", alg, lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, None, None))
    # the Winograd F(2x2,3x3) form of the same layers (decomposition codes 0x200 | w as the engine's plan picks them); algorithmic bytes
    # as for the direct form: the 16/9 larger transformed weight stream shows up as extra reads
    for (Cin, Cout, bs, code, name) in [(64, 64, 32, 0x204, "winograd layer1"), (128, 128, 16, 0x204, "winograd layer2"), (256, 256, 8, 0x207, "winograd layer3"),
                                        (512, 512, 4, 0x208, "winograd layer4"), (128, 128, 32, 0x204, "winograd up 1/4")]:
        feats = cl(torch.randn((n, Cin, bs, bs), device="cuda"))
        ring = torch.randn((128, Cin, 4 * bs), device="cuda")
        wpk = be.pack_conv3x3_weights(torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05)
        alg = 4.0 * (n * Cin * ((bs + 2) ** 2 + 4 * bs) + n * Cout * bs * bs + 9 * Cin * Cout)
        run(f"{name} ({n},{Cin}->{Cout},{bs}x{bs})", alg, lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, None, None, cfg=code))
    # the wide wave tile of the Winograd form (round 3, codes 0x400 | w), where the plan table picks it
    for (Cin, Cout, bs, code, name) in [(128, 128, 32, 0x401, "wide winograd up 1/4"), (128, 128, 16, 0x402, "wide winograd layer2"), (64, 64, 32, 0x400, "wide winograd layer1")]:
        feats = cl(torch.randn((n, Cin, bs, bs), device="cuda"))
        ring = torch.randn((128, Cin, 4 * bs), device="cuda")
        wpk = be.pack_conv3x3_weights(torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05)
        alg = 4.0 * (n * Cin * ((bs + 2) ** 2 + 4 * bs) + n * Cout * bs * bs + 9 * Cin * Cout)
        run(f"{name} ({n},{Cin}->{Cout},{bs}x{bs})", alg, lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, None, None, cfg=code))
    # shared-transform Winograd form (round 3, codes 0x200 | 11..13) at the shapes the refined plan table runs it, and the CSP head conv
    for (nn, Cin, Cout, bs, code, name) in [(64, 128, 128, 16, 0x20b, "shared-transform winograd layer2"), (64, 128, 128, 32, 0x20b, "shared-transform winograd up 1/4"),
                                            (64, 256, 256, 8, 0x20c, "shared-transform winograd layer3"), (38, 768, 256, 32, 0x20b, "shared-transform winograd CSP head"),
                                            (38, 768, 256, 32, 0x209, "winograd CSP head")]:
        gi2, m2 = grid_tables(1, 8, 16, nn)
        feats = cl(torch.randn((nn, Cin, bs, bs), device="cuda"))
        ring = torch.randn((128, Cin, 4 * bs), device="cuda")
        wpk = be.pack_conv3x3_weights(torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05)
        alg = 4.0 * (nn * Cin * ((bs + 2) ** 2 + 4 * bs) + nn * Cout * bs * bs + 9 * Cin * Cout)
        run(f"{name} ({nn},{Cin}->{Cout},{bs}x{bs})", alg, lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi2, m2, None, None, cfg=code))
    # Winograd F(4x4,3x3) (round 5, codes 0x1000 | c) at the shapes the plan table runs it: the 36/9 larger weight stream is extra reads
    for (nn, Cin, Cout, bs, code, name) in [(64, 64, 64, 32, 0x1000, "F(4x4) winograd layer1"), (64, 128, 128, 32, 0x1000, "F(4x4) winograd up 1/4"),
                                            (72, 128, 128, 16, 0x1000, "F(4x4) winograd layer2 (72 tiles)"), (38, 768, 256, 32, 0x1000, "F(4x4) winograd CSP head")]:
        gi2, m2 = grid_tables(1, 8, 16, nn)
        feats = cl(torch.randn((nn, Cin, bs, bs), device="cuda"))
        ring = torch.randn((128, Cin, 4 * bs), device="cuda")
        wpk = be.pack_conv3x3_weights(torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05)
        alg = 4.0 * (nn * Cin * ((bs + 2) ** 2 + 4 * bs) + nn * Cout * bs * bs + 9 * Cin * Cout)
        run(f"{name} ({nn},{Cin}->{Cout},{bs}x{bs})", alg, lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi2, m2, None, None, cfg=code))
    # the decoder's lateral 1x1 conv with the upsample term in its epilogue (round 5): skip + coarser map in, sum out
    for (nn, Cin, Cout, bs) in [(64, 64, 128, 32), (64, 128, 128, 16)]:
        x = cl(torch.randn((nn, Cin, bs, bs), device="cuda"))
        low = cl(torch.randn((nn, Cout, bs // 2, bs // 2), device="cuda"))
        wpk = be.pack_conv3x3_weights(torch.randn((Cout, Cin, 1, 1), device="cuda") * 0.05)
        alg = 4.0 * (nn * bs * bs * (Cin + Cout) + nn * (bs // 2) ** 2 * Cout + Cin * Cout)
        run(f"conv1x1 + upsample epilogue ({nn},{Cin}->{Cout},{bs}x{bs})", alg, lambda: be.conv1x1(x, wpk, Cout, None, None, upsample=(low, bs, False, 0.5, 0.5)))
    # dilation-2 form of the direct kernel (C5 backbone stage 4) and the dense prediction convs of the CSP head
    gi2, m2 = grid_tables(1, 8, 16, 38)
    feats = cl(torch.randn((38, 512, 8, 8), device="cuda"))
    ring = torch.randn((128, 512, 8 * 8), device="cuda")
    wpk = be.pack_conv3x3_weights(torch.randn((512, 512, 3, 3), device="cuda") * 0.05)
    run("conv3x3 dilation 2 (38,512->512,8x8)", 4.0 * (38 * 512 * ((8 + 4) ** 2 + 8 * 8) + 38 * 512 * 64 + 9 * 512 * 512),
        lambda: be.conv3x3_ring(feats, ring, wpk, 512, gi2, m2, None, None, dilation=2))
    xmap = cl(torch.randn((1, 256, 256, 512), device="cuda"))
    for cout in (1, 2):
        wp = be.pack_pred3x3_weights(torch.randn((cout, 256, 3, 3), device="cuda") * 0.05)
        bp = torch.randn(cout, device="cuda")
        run(f"pred3x3 (1,256,256,512) -> {cout}", 4.0 * 256 * 512 * (256 + cout), (lambda c_, w_, b_: lambda: be.pred3x3(xmap, w_, b_, c_))(cout, wp, bp))
    # network-input stage (window gather + 7x7 stem conv): frame-state windows in, packed stem output out
    state = torch.randn((1, 3, 1024, 2048), device="cuda")
    wst = be.pack_stem7x7_weights(cl(torch.randn((64, 3, 7, 7), device="cuda") * 0.05))
    run("stem7x7 (64 tiles of 128x128 -> 64x64x64)", 4.0 * (n * 3 * (128 + 6) ** 2 + n * 64 * 64 * 64 + 64 * 3 * 49),
        lambda: be.stem7x7(state, wst, m, 128, None))
    # the pyramid pooling of SwiftNet
    xp = cl(torch.randn((1, 128, 32, 64), device="cuda"))
    run("adaptive_avg_pool_nhwc (1,128,32,64) -> 8x16", 4.0 * (xp.numel() + 128 * 8 * 16), lambda: be.adaptive_avg_pool(xp, (8, 16)))
    # fused epilogue pass and per-tile bilinear resampling
    x = cl(torch.randn((n, 128, 16, 16), device="cuda"))
    sc = torch.rand(128, device="cuda")
    run("affine_act_nhwc (64,128,16,16) + residual", 3.0 * x.numel() * 4, lambda: be.affine_act(x, sc, sc, x, True))
    run("interp_bilinear_nhwc (64,128,16,16)->32x32 + skip add", (x.numel() + 2 * 4 * x.numel()) * 4.0,
        lambda: be.interp_bilinear(x, 32, 32, False, 0.5, 0.5, (None, None, cl(torch.zeros((n, 128, 32, 32), device="cuda")), False)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "pmc_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
