"""TinyCSP: a miniature, mmcv-free detector with the op classes of the reference's Pedestron CSP path that SwiftNet
does not exercise -- test DATA MODEL shared by oracle/gen_golden.py (where `blockcopy` is the REFERENCE package) and by
the parity tests (where `blockcopy` is this repo's):

* dilation-2 3x3 conv, padding 2  -> halo width 2            (reference backbones/resnet.py:155-162)
* ConvTranspose2d k4 s2 p1 and k4 s4 p0, run per tile, NO halo (reference necks/csp_neck.py:37-39, 68-83)
* L2Norm = pow/sum(dim=1)/sqrt/div on packed tiles            (necks/csp_neck.py:86-105)
* conv3x3 -> GroupNorm(32) -> ReLU on packed tiles            (batched trick, core/tensorwrapper.py:600-633)
* `blockcopy.to_tensor` inside the head, 3 branches           (anchor_heads/csp_head.py:135-151)
* three padded convs on the SAME packed tensor                (the head branches share their input)

Pure torch.nn; only the public `blockcopy` API is used (to_tensorwrapper / process_temporal_features / to_blocks /
combine_ / to_tensor), so the same source runs on both implementations."""
import torch
import torch.nn as nn


class TinyBackbone(nn.Module):
    """tile 32 -> 16 (stem s2) -> 8 (s2; C3) -> 4 (s2; C4) -> 4 (dilation 2; C5)."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Conv2d(3, 16, 3, stride=2, padding=1)
        self.l3 = nn.Conv2d(16, 32, 3, stride=2, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(32)
        self.l4 = nn.Conv2d(32, 48, 3, stride=2, padding=1, bias=False)
        self.bn4 = nn.BatchNorm2d(48)
        self.l5 = nn.Conv2d(48, 64, 3, stride=1, padding=2, dilation=2, bias=False)
        self.bn5 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.relu(self.stem(x))
        c3 = self.relu(self.bn3(self.l3(x)))
        c4 = self.relu(self.bn4(self.l4(c3)))
        c5 = self.relu(self.bn5(self.l5(c4)))
        return c3, c4, c5


class L2Norm(nn.Module):
    def __init__(self, n_channels, scale):
        super().__init__()
        self.eps = 1e-10
        self.weight = nn.Parameter(torch.full((n_channels,), float(scale)))

    def forward(self, x):
        norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + self.eps
        x = torch.div(x, norm)
        return self.weight.unsqueeze(0).unsqueeze(2).unsqueeze(3).expand_as(x) * x


class TinyNeck(nn.Module):
    def __init__(self):
        super().__init__()
        self.p3 = nn.ConvTranspose2d(32, 16, kernel_size=4, stride=2, padding=1)    # tile 8 -> 16
        self.p4 = nn.ConvTranspose2d(48, 16, kernel_size=4, stride=4, padding=0)    # tile 4 -> 16
        self.p5 = nn.ConvTranspose2d(64, 16, kernel_size=4, stride=4, padding=0)    # tile 4 -> 16
        self.p3_l2, self.p4_l2, self.p5_l2 = L2Norm(16, 10), L2Norm(16, 10), L2Norm(16, 10)

    def forward(self, inputs):
        p3 = self.p3_l2(self.p3(inputs[0]))
        p4 = self.p4_l2(self.p4(inputs[1]))
        p5 = self.p5_l2(self.p5(inputs[2]))
        return (torch.cat([p3, p4, p5], dim=1),)


class ConvGNReLU(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        self.gn = nn.GroupNorm(32, cout)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activate(self.gn(self.conv(x)))


class TinyHead(nn.Module):
    def __init__(self, in_channels=48, feat_channels=64):
        super().__init__()
        self.cls_convs = nn.ModuleList([ConvGNReLU(in_channels, feat_channels)])
        self.reg_convs = nn.ModuleList([ConvGNReLU(in_channels, feat_channels)])
        self.offset_convs = nn.ModuleList([ConvGNReLU(in_channels, feat_channels)])
        self.csp_cls = nn.Conv2d(feat_channels, 1, 3, padding=1)
        self.csp_reg = nn.Conv2d(feat_channels, 1, 3, padding=1)
        self.csp_offset = nn.Conv2d(feat_channels, 2, 3, padding=1)

    def forward(self, feats):
        import blockcopy

        x = feats[0]
        outs = []
        for convs in (self.cls_convs, self.reg_convs, self.offset_convs):
            f = x
            for layer in convs:
                f = blockcopy.to_tensor(layer(f))
            outs.append(f)
        return self.csp_cls(outs[0]), self.csp_reg(outs[1]).float(), self.csp_offset(outs[2]).float()


def tinycsp_arch():
    return TinyBackbone(), TinyNeck(), TinyHead()


def seed_weights(modules, prefixes=("backbone", "neck", "bbox_head")):
    """Name-seeded weights (bc_workloads.seeded) over '<prefix>.<key>' -- the keys the modules have inside a detector."""
    from bc_workloads import seeded

    for pre, m in zip(prefixes, modules):
        sd = m.state_dict()
        vals = seeded.name_seeded_state_dict({f"{pre}.{k}": v for k, v in sd.items()})
        m.load_state_dict({k: vals[f"{pre}.{k}"] for k in sd}, strict=True)
        m.eval()
    return modules


CFG = dict(N=1, H=128, W=192, block_size=32, n_frames=6, frame_seed0=31000, grid_seed=61)


def tinycsp_grids(cfg=CFG):
    """all, half, one tile, all-but-one, quarter, half -- seeded (bc_workloads.seeded.fixed_fraction_grid)."""
    from bc_workloads import seeded

    N, GH, GW = cfg["N"], cfg["H"] // cfg["block_size"], cfg["W"] // cfg["block_size"]
    total, s = N * GH * GW, cfg["grid_seed"]
    return [torch.ones(N, 1, GH, GW, dtype=torch.bool),
            seeded.fixed_fraction_grid(s + 1, N, GH, GW, total // 2),
            seeded.fixed_fraction_grid(s + 2, N, GH, GW, 1),
            seeded.fixed_fraction_grid(s + 3, N, GH, GW, total - 1),
            seeded.fixed_fraction_grid(s + 4, N, GH, GW, total // 4),
            seeded.fixed_fraction_grid(s + 5, N, GH, GW, total // 2)][:cfg["n_frames"]]


def run_detector_clip(modules, frames, grids, device="cpu"):
    """The reference's per-frame detector flow (detectors/csp_blockcopy.py:62-77) with a forced grid, written against the
    public `blockcopy` API only.  Returns per frame (head maps, frame_state)."""
    import blockcopy

    backbone, neck, head = modules
    feats_prev = None
    out = []
    with torch.no_grad():
        for img, grid in zip(frames, grids):
            x = blockcopy.to_tensorwrapper(img.to(device))
            feats_prev = x.process_temporal_features(feats_prev)
            x = x.to_blocks(grid.to(device))
            frame_state = x.combine_().to_tensor()
            maps = head(neck(backbone(x)))
            out.append(([m.detach().clone() for m in maps], frame_state.detach().clone()))
    return out
