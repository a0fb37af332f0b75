"""Information-gain rewards for the RL policy (reference: policy/information_gain.py:22-160)."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class InformationGain(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.num_classes = num_classes

    def get_output_repr(self, policy_meta: Dict) -> torch.Tensor:
        raise NotImplementedError

    def forward(self, policy_meta: Dict) -> torch.Tensor:
        raise NotImplementedError


class InformationGainSemSeg(InformationGain):
    """Per-pixel KL(prev || cur) of the quarter-scale softmaxed logits, averaged over classes (reference :22-41)."""

    def __init__(self, num_classes):
        super().__init__(num_classes)
        self.scale_factor = 1 / 4

    def get_output_repr(self, policy_meta: Dict) -> torch.Tensor:
        out = policy_meta["outputs"]
        assert out.size(1) == self.num_classes
        return out

    def forward(self, policy_meta: Dict) -> torch.Tensor:
        assert policy_meta["outputs"] is not None and policy_meta["outputs_prev"] is not None
        cur = F.log_softmax(F.interpolate(policy_meta["outputs"], scale_factor=self.scale_factor, mode="bilinear"), dim=1)
        prev = F.log_softmax(F.interpolate(policy_meta["outputs_prev"], scale_factor=self.scale_factor, mode="bilinear"), dim=1)
        # elementwise KL with log-space target: exp(prev) * (prev - cur)
        return F.kl_div(input=cur, target=prev, reduction="none", log_target=True).mean(1, keepdim=True)


def _as_int_boxes(arr: np.ndarray, div: int = 1) -> np.ndarray:
    return (arr[:, :4] / div).astype(np.int32)


def iou_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(n,4) x (m,4) integer boxes -> (n,m) float64 IoU with the reference's pixel convention (no +1, :121-160); a box
    that collapses at half resolution overlaps nothing (the reference asserts there, :137-140)."""
    a, b = a.astype(np.int64), b.astype(np.int64)
    ax1, ay1, ax2, ay2 = (a[:, None, i] for i in range(4))
    bx1, by1, bx2, by2 = (b[None, :, i] for i in range(4))
    xl, yt, xr, yb = np.maximum(ax1, bx1), np.maximum(ay1, by1), np.minimum(ax2, bx2), np.minimum(ay2, by2)
    valid = (ax1 < ax2) & (ay1 < ay2) & (bx1 < bx2) & (by1 < by2) & (xr >= xl) & (yb >= yt)
    inter = (xr - xl) * (yb - yt)
    union = (ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / union.astype(np.float64)
    return np.where(valid, iou, 0.0)


def paint_boxes_max(H: int, W: int, boxes: np.ndarray, values: np.ndarray, device="cpu", chunk: int = 0) -> torch.Tensor:
    """(H,W) float32 map = max over boxes k of values[k] inside [y1,y2) x [x1,x2), 0 elsewhere: the order-free form of the
    reference's sequence of ``mask[y1:y2, x1:x2] = max(mask[...], v)`` slices (values are >= 0), evaluated as a few batched
    tensor ops on ``device`` instead of one tiny launch per box."""
    out = torch.zeros((H, W), dtype=torch.float32, device=device)
    if len(boxes) == 0:
        return out
    if chunk <= 0:      # bound the transient (chunk, H, W) product to ~64 MB: 8 boxes at a time on a 1024x2048 map, 64 on small ones
        chunk = int(max(1, min(64, (16 << 20) // max(1, H * W))))
    assert (boxes >= 0).all(), "boxes must be clipped to the image"
    bt = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.int64)).to(device)
    vt = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64).astype(np.float32)).to(device)
    ys = torch.arange(H, device=device).view(1, H)
    xs = torch.arange(W, device=device).view(1, W)
    for k0 in range(0, len(boxes), chunk):
        bx, v = bt[k0:k0 + chunk], vt[k0:k0 + chunk]
        rows = ((ys >= bx[:, 1:2]) & (ys < bx[:, 3:4])).to(torch.float32)          # (K,H)
        cols = ((xs >= bx[:, 0:1]) & (xs < bx[:, 2:3])).to(torch.float32) * v.view(-1, 1)   # (K,W), value folded in
        out = torch.maximum(out, (rows.unsqueeze(2) * cols.unsqueeze(1)).amax(0))
    return out


def build_instance_mask(bbox_results, size, device="cpu") -> torch.Tensor:
    """Dense score mask of the detections of image 0 (output representation fed to the policy net; reference :55-66).
    Channel 0 is the running maximum over its boxes; further channels (unused by the shipped configs, which have one
    class) follow the reference's read-channel-0 / write-channel-c rule box by box."""
    mask = torch.zeros(size, device=device)
    if size[1] > 0:
        dets = bbox_results[0][0]
        mask[0, 0] = paint_boxes_max(size[2], size[3], _as_int_boxes(dets), dets[:, 4].astype(np.float64), device)
    for c in range(1, size[1]):
        dets = bbox_results[0][c]
        for (x1, y1, x2, y2), score in zip(_as_int_boxes(dets), dets[:, 4].tolist()):
            mask[0, c, y1:y2, x1:x2] = torch.clamp(mask[0, 0, y1:y2, x1:x2], min=score)
    return mask


def build_instance_mask_iou_gain(bbox_results, bbox_results_prev, size, device="cpu", SUBSAMPLE=2) -> torch.Tensor:
    """Detection information gain: (1 - IoU with the best-matching previous box) x score painted over both boxes;
    unmatched previous boxes paint their own score (reference :68-108, batch size 1).  Matching is one IoU matrix per
    class, painting one batched max-rasterisation -- no per-box Python loop, no per-box launch (the reference's loops
    bound C5 by a few ms per trained frame)."""
    assert len(bbox_results) == 1, "only supports batch size 1"
    H, W = size[2] // SUBSAMPLE, size[3] // SUBSAMPLE
    boxes, values = [], []
    for c in range(size[1]):
        cur, prev = bbox_results[0][c], bbox_results_prev[0][c]
        cb, pb = _as_int_boxes(cur, SUBSAMPLE), _as_int_boxes(prev, SUBSAMPLE)
        cs, ps = cur[:, 4].astype(np.float64), prev[:, 4].astype(np.float64)
        n, m = len(cb), len(pb)
        matched = np.zeros(m, dtype=bool)
        if n:
            if m:
                iou = iou_matrix(cb, pb)
                best_j = iou.argmax(axis=1)                     # first maximum, like the reference's strict '>' scan
                best = iou[np.arange(n), best_j]
                has = best > 0
            else:
                best_j, best, has = np.zeros(n, dtype=np.int64), np.zeros(n), np.zeros(n, dtype=bool)
            gain = 1.0 - np.where(has, best, 0.0)
            boxes.append(cb)
            values.append(gain * cs)
            if has.any():
                boxes.append(pb[best_j[has]])
                values.append(gain[has] * ps[best_j[has]])
                matched[best_j[has]] = True
        if m and not matched.all():
            boxes.append(pb[~matched])
            values.append(ps[~matched])
    mask = torch.zeros((size[0], size[1], H, W), device=device)
    if boxes:
        mask[0, 0] = paint_boxes_max(H, W, np.concatenate(boxes), np.concatenate(values), device)   # every class paints channel 0 (:93-104)
    if SUBSAMPLE > 1:
        mask = F.interpolate(mask, scale_factor=SUBSAMPLE, mode="nearest")
    return mask


class InformationGainObjectDetection(InformationGain):
    def get_output_repr(self, policy_meta: Dict) -> torch.Tensor:
        N, C, H, W = policy_meta["inputs"].shape
        return build_instance_mask(policy_meta["outputs"], (N, self.num_classes, H, W), device=policy_meta["inputs"].device)

    def forward(self, policy_meta: Dict) -> torch.Tensor:
        N, C, H, W = policy_meta["inputs"].shape
        return build_instance_mask_iou_gain(policy_meta["outputs"], policy_meta["outputs_prev"],
                                            (N, self.num_classes, H, W), device=policy_meta["inputs"].device)
