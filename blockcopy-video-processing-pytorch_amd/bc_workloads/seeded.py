"""Deterministic, name-seeded synthetic weights and frames.

No checkpoints or datasets exist offline (reference: semantic_segmentation/pretrained/README.md),
so every benchmark and parity test fills a model's ``state_dict`` from its key *names*: the
same keys give the same tensors in this container (where the golden fixtures are generated from
the reference model) and on the GPU box (where only this repo exists).  BN statistics are
randomised so that BN folding (reference: semantic_segmentation/lib/utils/bn_fusion.py:6-74)
is a non-trivial transformation.
"""
from __future__ import annotations

import math
import zlib

import torch


# 0.7 x Kaiming(fan_out) keeps SwiftNet-RN18 activations and logits O(1..10) with the randomised BN
# statistics below, so that an absolute 1e-4 logits tolerance is a meaningful fp32 bound.
CONV_GAIN = 0.7


def _gen(name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode("utf-8")))
    return g


def name_seeded_state_dict(template: dict) -> dict:
    """Return a new state_dict with the keys/shapes/dtypes of ``template`` and name-seeded values."""
    out = {}
    for key, ref in template.items():
        g = _gen(key)
        shape = tuple(ref.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=ref.dtype)
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) + 0.5
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "weight" and len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * (CONV_GAIN * math.sqrt(2.0 / fan_out))
        elif leaf == "weight" and len(shape) == 1:
            t = torch.rand(shape, generator=g) + 0.5
        else:
            t = torch.randn(shape, generator=g) * 0.05
        out[key] = t.to(ref.dtype)
    return out


def synthetic_frame(seed: int, shape, dtype=torch.float32) -> torch.Tensor:
    """``torch.randn`` frame from a CPU generator (device-independent values)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return torch.randn(tuple(shape), generator=g).to(dtype)


def fixed_fraction_grid(seed: int, N: int, GH: int, GW: int, n_exec: int) -> torch.Tensor:
    """Bool grid (N,1,GH,GW) with exactly ``n_exec`` executed tiles, chosen by a seeded
    ``randperm`` (the reference's `random` policy is unseeded, policy.py:210,142)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    total = N * GH * GW
    grid = torch.zeros(total, dtype=torch.bool)
    grid[torch.randperm(total, generator=g)[:n_exec]] = True
    return grid.view(N, 1, GH, GW)


# (weight gain, bias shift) of the CSP head's three prediction convs for the reference-pinned detector fixture
# (tests/golden/csp_ref_modules.npz): the name-seeded fill scales by fan_OUT, which for a 256 -> 1 conv gives logits of +-100
# (saturated scores, e^12-pixel boxes); with these the scores spread around the 0.1 threshold and boxes are ~30 px tall, so the
# top-k cut, the score threshold, NMS and the per-image cap of the decode all have work to do.
CSP_REF_OVERRIDES = {"csp_cls": (0.15, -4.5), "csp_reg": (0.03, 2.0), "csp_offset": (0.03, 0.0)}


def csp_reference_weights(template: dict) -> dict:
    """Name-seeded values over a CSP detector's state_dict keys + CSP_REF_OVERRIDES (shared by oracle/gen_golden.py, where the
    keys are the REFERENCE detector's, and by the parity tests, where they are this repo's: same names, same tensors)."""
    vals = name_seeded_state_dict(template)
    for k, (gain, bias) in CSP_REF_OVERRIDES.items():
        vals[f"bbox_head.{k}.weight"] = vals[f"bbox_head.{k}.weight"] * gain
        vals[f"bbox_head.{k}.bias"] = vals[f"bbox_head.{k}.bias"] + bias
    vals["bbox_head.reg_scales.0.scale"] = torch.tensor(1.0)
    vals["bbox_head.offset_scales.0.scale"] = torch.tensor(1.0)
    return vals
