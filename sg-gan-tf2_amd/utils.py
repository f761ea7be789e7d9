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
