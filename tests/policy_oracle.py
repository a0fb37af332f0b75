"""TEST INFRASTRUCTURE: loop restatements of the reference's detection reward functions
(blockcopy/blockcopy/policy/information_gain.py:55-160), box by box and slice by slice as the reference writes them.  The
product's vectorised forms (blockcopy.policy.information_gain) are checked bit for bit against these."""
import numpy as np
import torch
import torch.nn.functional as F


def _as_int_boxes(arr: np.ndarray, div: int = 1) -> np.ndarray:
    return (arr[:, :4] / div).astype(np.int32)


def box_iou(a, b) -> float:
    ax1, ay1, ax2, ay2 = a
    bx1, by1, bx2, by2 = b
    if not (ax1 < ax2 and ay1 < ay2 and bx1 < bx2 and by1 < by2):
        return 0.0   # a box that collapses at half resolution overlaps nothing (the reference asserts here, :137-140)
    xl, yt, xr, yb = max(ax1, bx1), max(ay1, by1), min(ax2, bx2), min(ay2, by2)
    if xr < xl or yb < yt:
        return 0.0
    inter = (xr - xl) * (yb - yt)
    return inter / float((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter)


def build_instance_mask(bbox_results, size, device="cpu") -> torch.Tensor:
    """Dense score mask of the detections of image 0 (output representation fed to the policy net)."""
    mask = torch.zeros(size, device=device)
    for c in range(size[1]):
        dets = bbox_results[0][c]
        for (x1, y1, x2, y2), score in zip(_as_int_boxes(dets), dets[:, 4].tolist()):
            mask[0, c, y1:y2, x1:x2] = torch.clamp(mask[0, 0, y1:y2, x1:x2], min=score)
    return mask


def build_instance_mask_iou_gain(bbox_results, bbox_results_prev, size, device="cpu", SUBSAMPLE=2) -> torch.Tensor:
    """Detection information gain: (1 - IoU with the best-matching previous box) x score painted over both boxes;
    unmatched previous boxes paint their own score (reference :68-108, batch size 1)."""
    assert len(bbox_results) == 1, "only supports batch size 1"
    mask = torch.zeros((size[0], size[1], size[2] // SUBSAMPLE, size[3] // SUBSAMPLE), device=device)

    def paint(box, value):
        x1, y1, x2, y2 = box
        mask[0, 0, y1:y2, x1:x2] = torch.clamp(mask[0, 0, y1:y2, x1:x2], min=float(value))

    for c in range(size[1]):
        cur, prev = bbox_results[0][c], bbox_results_prev[0][c]
        cur_boxes, prev_boxes = _as_int_boxes(cur, SUBSAMPLE), _as_int_boxes(prev, SUBSAMPLE)
        cur_scores, prev_scores = cur[:, 4].tolist(), prev[:, 4].tolist()
        matched = set()
        for box, score in zip(cur_boxes, cur_scores):
            best, best_j = 0.0, None
            for j, pbox in enumerate(prev_boxes):
                iou = box_iou(box, pbox)
                if iou > best:
                    best, best_j = iou, j
            matched.add(best_j)
            gain = 1.0 - best
            paint(box, gain * score)
            if best_j is not None:
                paint(prev_boxes[best_j], gain * prev_scores[best_j])
        for j, pbox in enumerate(prev_boxes):
            if j not in matched:
                paint(pbox, prev_scores[j])
    if SUBSAMPLE > 1:
        mask = F.interpolate(mask, scale_factor=SUBSAMPLE, mode="nearest")
    return mask
