"""A/B of the round-4 norm-statistics epilogues (Conv2DTranspose layers, stem) inside the cycle step, one process per arm:
    python tools/ab_deconv_stats.py <bits>     bit 0 = transposed layers, bit 1 = stem"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sggan_amd
from sggan_amd import kernels as K

on = int(sys.argv[1]) if len(sys.argv) > 1 else 3          # bit 0: the transposed layers' statistics epilogue, bit 1: the stem's
orig = K.ConvGeom.__init__
def patched(self, desc, x_shape, y_shape, dtype, is_deconv):   # a layer whose bit is off takes its separate statistics pass again
    orig(self, desc, x_shape, y_shape, dtype, is_deconv)
    if (is_deconv and not on & 1) or (not is_deconv and desc.R == 7 and not on & 2):
        self.stats_chunks = 0
K.ConvGeom.__init__ = patched
m = sggan_amd.sggan(sggan_amd.default_args(dtype="bf16", device="cuda:0", image_height=256, image_width=512, batch_size=8, cycle=True, graph=True,
                                           fuse_in_stats_deconv=True, fuse_in_stats_stem=True))      # (both on; the patch above takes an arm's layers off again)
bench.set_inputs(m, 8, 256, 512, 19)
for _ in range(5):
    m.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    m.train_step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{8 / dt:.1f} images/s  {dt * 1e3:.2f} ms/step  losses {m.losses()}")
