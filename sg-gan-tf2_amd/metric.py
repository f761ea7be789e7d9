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


def _fast_hist(label_true, label_pred, n_class):
    """metric.py:18-24 -- (n_class, n_class) int64 confusion matrix; labels as int arrays/tensors of equal size."""
    dev = "cuda"
    lt = torch.as_tensor(np.asarray(label_true) if not isinstance(label_true, torch.Tensor) else label_true).to(dev).to(torch.int32).contiguous().view(-1)
    lp = torch.as_tensor(np.asarray(label_pred) if not isinstance(label_pred, torch.Tensor) else label_pred).to(dev).to(torch.int32).contiguous().view(-1)
    assert lt.numel() == lp.numel()
    hist = torch.zeros(n_class * n_class, dtype=torch.int64, device=dev)
    A.check(A.lib().sgg_confusion_hist(K._p(lt), K._p(lp), lt.numel(), n_class, K._p(hist), K._s()), "confusion_hist")
    return hist.view(n_class, n_class).cpu().numpy()


def scores(label_trues, label_preds, n_class):
    """metric.py:27-47 -- same keys, same arithmetic on the accumulated histogram."""
    hist = np.zeros((n_class, n_class))
    for lt, lp in zip(label_trues, label_preds):
        hist += _fast_hist(lt, lp, n_class)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        valid = hist.sum(axis=1) > 0
        mean_iu = np.nanmean(iu[valid])
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return {"Overall Acc": acc, "Mean Acc": acc_cls, "FreqW Acc": fwavacc, "Mean IoU": mean_iu,
            "Class IoU": dict(zip(range(n_class), iu))}


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
