"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the confusion-matrix metrics of the reference's
StreamMetrics (metrics/stream_metrics.py:24-63).  The reference module cannot be imported here (its package
needs cv2), so this follows the source text; pinned by a hand-worked case in
tests/test_oracle_golden.py.  Never imported by the product."""
import numpy as np


def fast_hist(label_true, label_pred, n_classes):
    """reference :24-31"""
    label_true = np.asarray(label_true).reshape(-1)
    label_pred = np.asarray(label_pred).reshape(-1)
    mask = (label_true >= 0) & (label_true < n_classes)
    return np.bincount(n_classes * label_true[mask].astype(int) + label_pred[mask],
                       minlength=n_classes ** 2).reshape(n_classes, n_classes)


def foreground_metrics(hist, fg=1):
    """reference :33-63 -> (miou, foreground_iou, precision, recall, f1)"""
    hist = np.asarray(hist, dtype=np.float64)
    tp = hist[fg, fg]
    fp = hist[:, fg].sum() - tp
    fn = hist[fg, :].sum() - tp
    eps = 1e-7
    fiou = tp / (tp + fp + fn + eps)
    precision = tp / (tp + fp + eps)
    recall = tp / (tp + fn + eps)
    f1 = 2 * precision * recall / (precision + recall + eps)
    btp = hist[0, 0]
    bfp = hist[:, 0].sum() - btp
    bfn = hist[0, :].sum() - btp
    biou = btp / (btp + bfp + bfn + eps)
    return (biou + fiou) / 2.0, fiou, precision, recall, f1
