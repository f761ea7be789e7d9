"""Evaluation metrics -- mirror of the NumPy part of the reference's ``metric.py`` (next-row SURVEY.md 8(f)4).

``_fast_hist`` / ``scores`` (metric.py:18-47) with the confusion matrix accumulated on the GPU (integer atomics:
exact and order-independent) and ``scores_seg_fake`` (metric.py:71-77).  ``dense_crf`` (pydensecrf) is out of scope.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _abi as A
from . import kernels as K


def _labels(x, dev):
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    return t.to(dev).to(torch.int32).contiguous().view(-1)


def _accumulate_hist(hist, label_true, label_pred, n_class):
    """Adds one (true, predicted) label pair to the device-resident int64 confusion matrix ``hist`` (flat n_class^2)."""
    lt, lp = _labels(label_true, hist.device), _labels(label_pred, hist.device)
    assert lt.numel() == lp.numel()
    A.check(A.lib().sgg_confusion_hist(K._p(lt), K._p(lp), lt.numel(), n_class, K._p(hist), K._s()), "confusion_hist")


def _fast_hist(label_true, label_pred, n_class):
    """metric.py:18-24 -- (n_class, n_class) int64 confusion matrix; labels as int arrays/tensors of equal size."""
    hist = torch.zeros(n_class * n_class, dtype=torch.int64, device="cuda")
    _accumulate_hist(hist, label_true, label_pred, n_class)
    return hist.view(n_class, n_class).cpu().numpy()


def scores(label_trues, label_preds, n_class):
    """metric.py:27-47 -- the FCN score set (same keys).  The confusion matrix of ALL pairs is accumulated on the device
    in one int64 buffer (exact, order independent) and read back once; the five scores are then ratios of its
    per-class true-positive / ground-truth / prediction counts."""
    hist = torch.zeros(n_class * n_class, dtype=torch.int64, device="cuda")
    for lt, lp in zip(label_trues, label_preds):
        _accumulate_hist(hist, lt, lp, n_class)
    h = hist.view(n_class, n_class).cpu().numpy()
    tp = np.diagonal(h).astype(np.float64)            # correctly labelled pixels per class
    n_true = h.sum(axis=1).astype(np.float64)         # pixels whose ground truth is the class
    n_pred = h.sum(axis=0).astype(np.float64)         # pixels predicted as the class
    total = float(h.sum())
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = tp / (n_true + n_pred - tp)
        seen = n_true > 0
        share = n_true / total
        out = {"Overall Acc": tp.sum() / total,
               "Mean Acc": np.nanmean(tp / n_true),
               "FreqW Acc": (share[share > 0] * iou[share > 0]).sum(),
               "Mean IoU": np.nanmean(iou[seen]),
               "Class IoU": {c: iou[c] for c in range(n_class)}}
    return out


def argmax_u8_labels(img, c_real=None):
    """labels (N,H,W) int32 = argmax over channels of uint8(255*img) -- the label rule of scores_seg_fake.
    img: (N,H,W,C) numpy / torch (float32 real channels, or an internal channel-padded activation with c_real given)."""
    t = img if isinstance(img, torch.Tensor) else torch.as_tensor(np.asarray(img, dtype=np.float32))
    t = t.cuda().contiguous()
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    Cp = t.shape[-1]
    Cr = Cp if c_real is None else c_real
    out = torch.empty(t.shape[:-1], dtype=torch.int32, device=t.device)
    A.check(A.lib().sgg_argmax_u8_labels(K._p(t), K._p(out), out.numel(), Cr, Cp, K.dt(t), K._s()), "argmax_u8_labels")
    return out


def scores_seg_fake(seg_image, fake_img):
    """metric.py:71-77: true labels from seg_image, predicted labels from fake_img; both returned transposed to
    (N, W, H) exactly as ``np.argmax(x.transpose(0,3,2,1), axis=1)`` does."""
    f = fake_img.tensor() if hasattr(fake_img, "tensor") else fake_img
    gts = argmax_u8_labels(seg_image).permute(0, 2, 1).contiguous()
    preds = argmax_u8_labels(f).permute(0, 2, 1).contiguous()
    return gts.cpu().numpy(), preds.cpu().numpy()
