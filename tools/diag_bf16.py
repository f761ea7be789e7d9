"""Diagnostic: per-parameter agreement of the bf16 step's gradients with the float64 oracle fixture."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sggan_amd
from tests.test_gpu_step import small_model

for dtype in ("f32", "bf16"):
    m, z = small_model(sggan_amd, dtype)
    m.train_step()
    print(dtype, "losses", m.losses(), float(z["gen_loss"]), float(z["disc_loss"]))
    for net, pre in ((m.generator, "gG/"), (m.discriminator, "gD/")):
        g = net.P.export(net.P.grad)
        for k, v in g.items():
            e = z[pre + k].astype(np.float64)
            if np.abs(e).max() < 1e-9:
                continue
            v = v.astype(np.float64)
            cos = (v * e).sum() / (np.linalg.norm(v) * np.linalg.norm(e) + 1e-30)
            print(f"  {pre}{k:10s} cos {cos:.5f}  relmax {np.abs(v - e).max() / np.abs(e).max():.4f}  norm ratio {np.linalg.norm(v) / np.linalg.norm(e):.4f}")
