"""Device-side counterpart of the confusion-matrix half of the reference's StreamMetrics
(metrics/stream_metrics.py:7-63,100-186): the 2x2 (n x n) histogram of (ground truth, prediction) is counted by
a HIP kernel straight from the label tensor and the argmax mask -- or from the logits, fusing the argmax
(train.py:644,659) -- and stays on the GPU as int64; only the n*n counts cross PCIe when results are read.

Kept from the reference: constructor arguments, `n_classes`, `confusion_matrix`, `FOREGROUND_CLASS`,
`_fast_hist`, `update(label_trues, label_preds, sequence_data)`, `get_results()` keys "MIoU", "Foreground IoU",
"Foreground F1", "Precision", "Recall" with the same formulas (eps = 1e-7), `reset()`.
Not built (they stay CPU/cv2/scipy code outside the hot path): the temporal, region and
front-tracking evaluators and the weighted "Best Score" that mixes them in.
"""
import numpy as np
import torch

from .. import ops


class StreamMetrics(object):
    def __init__(self, n_classes, sequence_length=7, temporal_stride=1, threshold=0.005, device=None):
        self.n_classes = n_classes
        self.FOREGROUND_CLASS = 1
        self.sequence_length, self.temporal_stride, self.threshold = sequence_length, temporal_stride, threshold
        self.device = torch.device(device if device is not None else "cuda")
        self._hist = torch.zeros((n_classes, n_classes), dtype=torch.int64, device=self.device)

    # ---- counting -------------------------------------------------------------------------------------
    def _dev(self, a, what):
        t = a if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a))
        if t.dtype not in (torch.uint8, torch.int64):
            if t.is_floating_point():
                raise TypeError("%s must be an integer class map, got %s" % (what, t.dtype))
            t = t.to(torch.int64)
        return t.to(self.device).contiguous()

    def _fast_hist(self, label_true, label_pred):
        """n x n int64 histogram (device tensor) of one batch: reference :24-31"""
        return ops.confusion_matrix(self._dev(label_true, "label_true"), self._dev(label_pred, "label_pred"),
                                    self.n_classes)

    def update(self, label_trues, label_preds, sequence_data=True):
        """reference :100-122: a sequence contributes its LAST frame, a batch contributes all of it"""
        if sequence_data:
            label_trues, label_preds = label_trues[-1], label_preds[-1]
        ops.confusion_matrix(self._dev(label_trues, "label_trues"), self._dev(label_preds, "label_preds"),
                             self.n_classes, hist=self._hist)

    def update_logits(self, label_trues, logits):
        """prediction = logits.max(1)[1] fused into the counting kernel; logits [B, C, H, W] fp32 on the device"""
        ops.confusion_matrix_logits(self._dev(label_trues, "label_trues"), logits, self.n_classes, hist=self._hist)

    # ---- results ---------------------------------------------------------------------------------------
    @property
    def confusion_matrix(self):
        """host copy, float64 like the reference's np.zeros accumulator (:12)"""
        return self._hist.cpu().numpy().astype(np.float64)

    def _calculate_foreground_metrics(self, hist):
        """reference :33-63 (without its debug prints)"""
        fg = self.FOREGROUND_CLASS
        true_positives = hist[fg, fg]
        false_positives = hist[:, fg].sum() - true_positives
        false_negatives = hist[fg, :].sum() - true_positives
        eps = 1e-7
        foreground_iou = true_positives / (true_positives + false_positives + false_negatives + eps)
        precision = true_positives / (true_positives + false_positives + eps)
        recall = true_positives / (true_positives + false_negatives + eps)
        f1_score = 2 * precision * recall / (precision + recall + eps)
        background_tp = hist[0, 0]
        background_fp = hist[:, 0].sum() - background_tp
        background_fn = hist[0, :].sum() - background_tp
        background_iou = background_tp / (background_tp + background_fp + background_fn + eps)
        miou = (background_iou + foreground_iou) / 2.0
        return miou, foreground_iou, precision, recall, f1_score

    def get_results(self, update_best=True):
        miou, foreground_iou, precision, recall, f1_score = self._calculate_foreground_metrics(self.confusion_matrix)
        return {"MIoU": miou, "Foreground IoU": foreground_iou, "Foreground F1": f1_score,
                "Precision": precision, "Recall": recall}

    def reset(self):
        self._hist.zero_()
