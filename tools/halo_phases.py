"""Where a tile of the residual-conv halo GEMM spends its cycles (lab build, SGG_ABLATE=8): block 0 stamps the shader clock at
six points of 16 mid-loop tiles, per wave.   python sg-gan-tf2_amd/build.py --lab && SGG_ABLATE=8 python tools/halo_phases.py [fwd|dgrad]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("SGG_ABLATE") == "8", "run with SGG_ABLATE=8"
import numpy as np
import torch
import labenv; labenv.select("libsggan_lab.so")
import sggan_amd
from sggan_amd import kernels as K, _abi as A

op = sys.argv[1] if len(sys.argv) > 1 else "fwd"
g = K.conv_geom(8, 64, 128, 256, 256, 3, 3, 1, "VALID", 1, torch.bfloat16)
x = torch.randn(g.x_shape, device="cuda").to(torch.bfloat16)
w = torch.randn((3, 3, 256, 256), device="cuda") / 48.0
wf, wd = K.pack_weights(w, 256, 256, torch.bfloat16)
L = C.CDLL(A.LIB_PATH)
for it in range(40):
    K.conv_fwd(g, x, wf, None) if op == "fwd" else K.conv_dgrad(g, x, wd)
torch.cuda.synchronize()
out = (C.c_ulonglong * 768)()
assert L.sgg_debug_phases(out) == 0
p = np.array(list(out), dtype=np.int64).reshape(8, 16, 6)
names = ["DMA issue", "frag loads + 64 MFMAs", "wait DMA landed (vmcnt 0)", "wait LDS (lgkmcnt 0)", "barrier"]
d = np.diff(p, axis=2)                       # (wave, tile, 5 phases)
tile = p[:, 1:, 0] - p[:, :-1, 0]            # loop top to loop top
print(f"{op}: cycles per tile (loop top to loop top), mean over 15 tiles, per wave:", tile.mean(1).round().astype(int).tolist())
for i, n in enumerate(names):
    print(f"  {n:28s} per wave: {d[:, :, i].mean(1).round().astype(int).tolist()}   all: {d[:, :, i].mean():.0f}")
print("  per-tile detail, wave 0:", d[0].tolist()[:6])
print("  per-tile detail, wave 4:", d[4].tolist()[:6])
