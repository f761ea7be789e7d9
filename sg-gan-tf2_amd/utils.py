"""Caller-side pieces next to the step -- mirror of the parts of the reference's ``utils.py`` that touch the hot
path (next-row SURVEY.md 8(f)2): ``ImagePool`` (utils.py:27-53) kept in device memory, and ``one_hot`` + the mask
resample (utils.py:158-165,197-199) as one GPU kernel (``segment_class.one_hot_mask``)."""
from __future__ import annotations

import numpy as np
import torch

from .segment_class import one_hot_mask  # noqa: F401  (utils.one_hot + zoom, fused)


class ImagePool(object):
    """History of generated images for the discriminators (utils.py:27-53), device resident.

    Same call protocol and decisions as the reference: ``pool([fake_A, fake_B, mask_A, mask_B])`` returns the input
    until ``maxsize`` entries are stored; afterwards, with probability 1/2, it swaps (fake_A, mask_A) with a random
    stored entry and (fake_B, mask_B) with another random stored entry and returns the old ones.  Decisions come from
    ``rng`` (a ``numpy.random.Generator``/``RandomState``-like with ``.rand()``; default = ``np.random`` like the
    reference, which leaves it unseeded).  Stored tensors never leave HBM."""

    def __init__(self, maxsize=50, rng=None):
        self.maxsize = maxsize
        self.num_img = 0
        self.images = []
        self._rng = rng if rng is not None else np.random

    def _rand(self):
        r = self._rng
        return float(r.rand()) if hasattr(r, "rand") else float(r.random())

    def __call__(self, image):
        if self.maxsize <= 0:                                  # utils.py:34-35
            return image
        if self.num_img < self.maxsize:                        # :36-39
            self.images.append([t.clone() for t in image])
            self.num_img += 1
            return image
        if self._rand() > 0.5:                                 # :40-51
            idx = int(self._rand() * self.maxsize)
            tmp1, tmp3 = self.images[idx][0], self.images[idx][2]
            self.images[idx][0], self.images[idx][2] = image[0].clone(), image[2].clone()
            idx = int(self._rand() * self.maxsize)
            tmp2, tmp4 = self.images[idx][1], self.images[idx][3]
            self.images[idx][1], self.images[idx][3] = image[1].clone(), image[3].clone()
            return [tmp1, tmp2, tmp3, tmp4]
        return image                                           # :52-53


class StaticImagePool(ImagePool):
    """``ImagePool`` with a FIXED launch sequence, so that the pool step can be replayed from a captured HIP graph
    (``sggan(use_pool=True, graph=True)``).  Same protocol and the same decisions from the same random stream as utils.py:27-53;
    what differs is where they act: the host only draws the decisions (``decide()``: four slot indices, written into a small
    device tensor BEFORE the step / the replay), and the step always launches the same three kernels per tensor --

        store[CUR] <- current tensor;   returned <- store[src] (a gather: the current one, or an older entry);   store[dst] <- current

    -- with src = dst = CUR when the protocol returns its input unchanged, dst = the fill slot while the pool fills, and
    src = dst = the random slot on a swap.  The history lives in four rings of ``maxsize`` + 1 slots in HBM, allocated at the
    first call.  The (fake_A, mask) and (fake_B, mask) halves of an entry are swapped independently by the reference
    (two random indices), so the rings are independent."""

    def __init__(self, maxsize=50, rng=None):
        super().__init__(maxsize, rng)
        self.store = None
        self.ctrl = None          # device int64 [4]: src_A, dst_A, src_B, dst_B
        self.last = None          # the host copy of the last decision (tests)

    def decide(self):
        """utils.py:34-53, decisions only.  Returns (src_A, dst_A, src_B, dst_B); CUR = ``maxsize`` is the slot of the current tensors."""
        cur = max(self.maxsize, 0)
        if self.maxsize <= 0:
            d = (cur, cur, cur, cur)
        elif self.num_img < self.maxsize:
            d = (cur, self.num_img, cur, self.num_img)
            self.num_img += 1
        elif self._rand() > 0.5:
            i1 = int(self._rand() * self.maxsize)
            i2 = int(self._rand() * self.maxsize)
            d = (i1, i1, i2, i2)
        else:
            d = (cur, cur, cur, cur)
        self.last = d
        return d

    def stage(self, device):
        """Draw this step's decisions and put them where the (recorded) step reads them.  Host side: call once per step, before it."""
        d = self.decide()
        if self.ctrl is None:
            self.ctrl = torch.zeros(4, dtype=torch.int64, device=device)
        # four scalar fills (kernel launches with an immediate argument): stream-ordered and asynchronous -- a host-to-device copy
        # of a pageable tensor would block the host until everything queued before it has run, once per step
        for k, v in enumerate(d):
            self.ctrl[k:k + 1].fill_(int(v))
        return d

    def __call__(self, image):
        """[fake_A, fake_B, mask_of_A, mask_of_B] -> the same four, current or from the history (always fresh tensors)."""
        assert self.ctrl is not None, "StaticImagePool.stage() must run before the step"
        slots = max(self.maxsize, 0) + 1
        if self.store is None or any(tuple(st.shape[1:]) != tuple(t.shape) or st.dtype != t.dtype for st, t in zip(self.store, image)):
            self.store = [torch.zeros((slots,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in image]
        out = []
        for k, (st, t) in enumerate(zip(self.store, image)):
            src, dst = (self.ctrl[0:1], self.ctrl[1:2]) if k in (0, 2) else (self.ctrl[2:3], self.ctrl[3:4])
            st[slots - 1].copy_(t)
            out.append(torch.index_select(st, 0, src)[0])
            st.index_copy_(0, dst, t.unsqueeze(0))
        return out


# ----------------------------------------------------------------------------- image helpers of the test paths
def inverse_transform(images):
    """utils.py:300-312: tanh range [-1,1] -> uint8 [0,255] (C cast: truncation)."""
    a = images.tensor() if hasattr(images, "tensor") else images
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float32)
    return np.array(((a + 1.) / 2) * 255).astype(np.uint8)


def merge(images, size):
    """utils.py:261-269: image k of a (B,h,w,3) batch goes to cell (k // size[1], k % size[1]) of a size[0] x size[1] grid;
    cells without an image stay black.  One reshape/transpose of the padded batch instead of a paste loop."""
    rows, cols = int(size[0]), int(size[1])
    images = np.asarray(images)
    b, h, w = images.shape[:3]
    if b > rows * cols:            # the reference's paste loop fails (broadcast error) on the first image outside the grid
        raise ValueError(f"merge: {b} images do not fit a {rows} x {cols} grid")
    cells = np.zeros((rows * cols, h, w, 3), dtype=np.float64)
    cells[:b] = images
    return cells.reshape(rows, cols, h, w, 3).transpose(0, 2, 1, 3, 4).reshape(rows * h, cols * w, 3).astype(np.uint8)


def get_img(image, size):
    """utils.py:243-247: (1, H*size0, W*size1, 3) uint8 array of the tiled, inverse-transformed images."""
    img = merge(inverse_transform(image), size)
    return img.reshape((1,) + img.shape)


def save_images(images, size, image_path):
    """utils.py:239-241 (imsave of the merged inverse transform); PNG/JPEG writing is host-side I/O (PIL)."""
    from PIL import Image
    Image.fromarray(merge(inverse_transform(images), size)).save(image_path)


def convert_image_dtype_uint8(sample):
    """tf.image.convert_image_dtype(float image in [0,1], uint8) as model.py:352,551 uses it: [3P] scale by dtype.max + 0.5 and
    cast (truncate); returned as float32 like the reference's ``np.array(...).astype(np.float32)`` -- i.e. the test paths feed
    the generator 0..255 values (the training loop feeds [0,1]; that is the reference's behaviour, reproduced, not endorsed)."""
    a = np.asarray(sample, dtype=np.float32)
    return (a * np.float32(255.5)).astype(np.uint8).astype(np.float32)


class SummarySink:
    """Scalar summaries of the epoch loop (model.py:263-268, 389-393: tf.summary.scalar under a train writer).  TensorFlow's
    event-file writer is not part of this build; the sink keeps the reference's scalar NAMES and steps, in memory and (optionally)
    as JSON lines -- one {"tag", "step", "value"} object per call -- which any tfevents writer can replay."""

    def __init__(self, path=None):
        self.path, self.records = path, []
        if path:
            import os
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def scalar(self, tag, value, step):
        rec = {"tag": tag, "step": int(step), "value": float(value)}
        self.records.append(rec)
        if self.path:
            import json
            with open(self.path, "a") as f:
                f.write(json.dumps(rec) + "\n")

    def image(self, tag, array, step):
        self.records.append({"tag": tag, "step": int(step), "image_shape": tuple(np.asarray(array).shape)})
