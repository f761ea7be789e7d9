"""Shader clock held by the residual-conv forward kernel while it runs (debug aid): the kernel timestamps one block with the
shader-clock counter and the constant-rate wall clock (SGG_ABLATE=9), so cycles / wall time = the clock under load.
    python sg-gan-tf2_amd/build.py --lab && SGG_ABLATE=9 python tools/shader_clock.py      (lab build: libsggan_lab.so)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("SGG_ABLATE") == "9", "run with SGG_ABLATE=9"
import torch
import labenv; labenv.select("libsggan_lab.so")
import sggan_amd
from sggan_amd import kernels as K, _abi as A

g = K.conv_geom(8, 64, 128, 256, 256, 3, 3, 1, "VALID", 1, torch.bfloat16)
x = torch.randn(g.x_shape, device="cuda").to(torch.bfloat16)
w = torch.randn((3, 3, 256, 256), device="cuda") / 48.0
wf, _ = K.pack_weights(w, 256, 256, torch.bfloat16)
L = C.CDLL(A.LIB_PATH)
out = (C.c_ulonglong * 5)()
for it in range(60):                      # sustained load first, then read the last launch's stamps
    K.conv_fwd(g, x, wf, None)
torch.cuda.synchronize()
assert L.sgg_debug_clocks(out) == 0
cyc, wall, khz = out[2] - out[0], out[3] - out[1], out[4]
us = wall / (khz * 1e3) * 1e6
print(f"block 0: {cyc} shader cycles in {wall} wall ticks at {khz} kHz = {us:.1f} us  ->  {cyc / us / 1e3:.3f} GHz under load")
