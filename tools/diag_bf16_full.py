"""Diagnostic: agreement of the bf16 path's gradients with the f32 path on the FULL-SIZE networks (same parameters,
same 1x256x512 batch -- the smallest input whose residual maps, 64x128, take the LDS-resident 3x3 kernels): cosine similarity and norm ratio per parameter tensor."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sggan_amd

g = torch.Generator().manual_seed(5)
real, seg = torch.rand((1, 256, 512, 3), generator=g), torch.rand((1, 256, 512, 3), generator=g)
mask = torch.nn.functional.one_hot(torch.randint(0, 34, (1, 5, 13), generator=g), 34).float()
res = {}
for dt in ("f32", "bf16", "mixed"):
    m = sggan_amd.sggan(sggan_amd.default_args(dtype="bf16" if dt == "mixed" else dt, mixed=(dt == "mixed"), seed=19))
    m.real_A, m.seg_A, m.mask_A = real, seg, mask
    m.train_step()
    res[dt] = (m.generator.P.export(m.generator.P.grad), m.discriminator.P.export(m.discriminator.P.grad), m.losses())
print("losses f32", res["f32"][2], "bf16", res["bf16"][2], "mixed", res["mixed"][2])
for mode in ("bf16", "mixed"):
  for idx, net in ((0, "G"), (1, "D")):
    worst = []
    for k, e in res["f32"][idx].items():
        v = res[mode][idx][k].astype(np.float64); e = e.astype(np.float64)
        if np.abs(e).max() < 1e-12:
            continue
        cos = (v * e).sum() / (np.linalg.norm(v) * np.linalg.norm(e) + 1e-30)
        worst.append((cos, k, np.linalg.norm(v) / np.linalg.norm(e)))
    worst.sort()
    print(mode, net, "min cos:", [(round(c, 4), k, round(r, 3)) for c, k, r in worst[:6]], " median cos:", round(float(np.median([c for c, _, _ in worst])), 5))
