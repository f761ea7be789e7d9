"""Colour -> class-index masks on the GPU -- mirror of the reference's ``segment_class.py``.

``cityscape()`` / ``A_maskmap()`` return the same 21-entry (R,G,B) -> class dictionary
(segment_class.py:60-73; read back from the kernel's own table so the two cannot drift);
``preprocess(img)`` is the per-pixel lookup of segment_class.py:87-97 as one HIP kernel
(bit exact, default class 0, alpha ignored).  PNG I/O and the process pool
(segment_class.py:79-99) are host tooling and out of scope.
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict

import numpy as np
import torch

from . import _abi as A
from . import kernels as K

num_seg_masks = 8       # segment_class.py:10


def cityscape():
    """segment_class.py:60-70."""
    keys = (C.c_uint32 * 32)()
    vals = (C.c_uint8 * 32)()
    n = A.lib().sgg_seg_class_table(keys, vals, 32)
    if n <= 0:
        raise A.SggError("sgg_seg_class_table failed")
    m = defaultdict(int)
    for i in range(n):
        k = int(keys[i])
        m[((k >> 16) & 255, (k >> 8) & 255, k & 255)] = int(vals[i])
    return m


def A_maskmap():
    """segment_class.py:72-73."""
    return cityscape()


def preprocess(img, device="cuda"):
    """segment_class.py:87-97: uint8 (M,N,3|4) [or a batch (B,M,N,3|4)] -> uint8 class indices, on the GPU.
    Accepts numpy or torch; returns a torch uint8 tensor on `device`."""
    t = img if isinstance(img, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(img))
    if t.dtype != torch.uint8 or t.shape[-1] < 3:
        raise ValueError("expected uint8 image(s) with >= 3 channels")
    return K.seg_class_map(t.to(device).contiguous())


def one_hot_mask(seg_class, out_h, out_w, num_classes=num_seg_masks):
    """utils.py:158-165 (one_hot) + utils.py:197-199 (zoom to the mask grid), fused: uint8 (B,M,N) class
    indices -> float32 (B,out_h,out_w,num_classes) one-hot of the align-corners nearest resample
    (deviation D1)."""
    t = seg_class if isinstance(seg_class, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(seg_class))
    if t.dim() == 2:
        t = t[None]
    return K.onehot_resample(t.to(torch.uint8).cuda().contiguous(), int(out_h), int(out_w), int(num_classes))
