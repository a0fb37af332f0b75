"""Segmentation quality metrics on the output side of the hot path: streaming confusion
matrix -> overall / mean accuracy, frequency-weighted accuracy, mean IoU, per-class IoU and the mIoU over the thin
classes.  Restates the reference's ``StreamSegMetrics`` (semantic_segmentation/lib/utils/metrics.py:25-118): same update
rule (labels outside [0, n_classes) are ignored), same result keys, same running per-image lists."""
from __future__ import annotations

import numpy as np

from .cityscapes import FINE_CLASSES, train_id_names


class StreamSegMetrics:
    def __init__(self, n_classes: int, single_iou_class: int = -1, classes=None):
        self.n_classes = n_classes
        self.confusion_matrix = np.zeros((n_classes, n_classes))
        self.single_iou_class = single_iou_class
        self.accs, self.accs_sum, self.ious, self.ious_sum = [], [], [], []
        self.classes = classes

    def _fast_hist(self, label_true, label_pred):
        keep = (label_true >= 0) & (label_true < self.n_classes)
        return np.bincount(self.n_classes * label_true[keep].astype(int) + label_pred[keep],
                           minlength=self.n_classes ** 2).reshape(self.n_classes, self.n_classes)

    def _miou(self, hist):
        with np.errstate(divide="ignore", invalid="ignore"):
            iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        return iu, (np.nanmean(iu) if self.single_iou_class < 0 else iu[self.single_iou_class])

    def update(self, label_trues, label_preds):
        """One (ground truth, prediction) pair per image; also records the per-image and running mIoU / accuracy."""
        for lt, lp in zip(label_trues, label_preds):
            h = self._fast_hist(np.asarray(lt).flatten(), np.asarray(lp).flatten())
            self.confusion_matrix += h
            with np.errstate(divide="ignore", invalid="ignore"):
                self.ious.append(self._miou(h)[1])
                self.accs.append(np.diag(h).sum() / h.sum())
                self.ious_sum.append(self._miou(self.confusion_matrix)[1])
                self.accs_sum.append(np.diag(self.confusion_matrix).sum() / self.confusion_matrix.sum())

    def get_results(self, per_class: bool = False):
        hist = self.confusion_matrix
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = np.diag(hist).sum() / hist.sum()
            acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
            iu, mean_iu = self._miou(hist)
            freq = hist.sum(axis=1) / hist.sum()
            fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
            fine_iu = np.nanmean(iu[self.classes]) if self.classes is not None else 0
        return {"Overall Acc": acc, "Mean Acc": acc_cls, "FreqW Acc": fwavacc, "Mean IoU": mean_iu,
                "Class IoU": dict(zip(train_id_names(), iu)), "Fine mIoU": fine_iu}

    @staticmethod
    def to_str(results):
        return "\n" + "".join("%s: %f\n" % (k, v) for k, v in results.items() if k != "Class IoU")

    def reset(self):
        self.confusion_matrix = np.zeros((self.n_classes, self.n_classes))


def cityscapes_metrics() -> StreamSegMetrics:
    """The reference driver's configuration (test_swiftnet.py:135): 19 classes, Fine mIoU over the thin classes."""
    return StreamSegMetrics(19, classes=FINE_CLASSES)
