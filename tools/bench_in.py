"""Micro-benchmark of instance-norm forward/backward at the bench shapes; GB/s against algorithmic bytes
(fwd: 2 reads + 1 write; bwd: 4 reads + 1 write of the tensor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import labenv; labenv.select()
import sggan_amd
from sggan_amd import kernels as K
from sggan_amd import _abi as A


def timeit(f, iters=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


SHAPES = ((8, 64, 128, 256), (16, 64, 128, 256), (8, 128, 256, 128), (8, 256, 512, 64), (8, 32, 64, 256), (8, 5, 13, 512))
if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "res":
    SHAPES = SHAPES[:1]                       # the residual-block tensor only (PMC passes: tools/pmc_in.sh)
DIR = sys.argv[sys.argv.index("--dir") + 1] if "--dir" in sys.argv else "both"
for shape in SHAPES:
    x = torch.randn(shape, device="cuda").to(torch.bfloat16)
    dy = torch.randn(shape, device="cuda").to(torch.bfloat16)
    C = shape[-1]
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    y, st = K.instnorm_fwd(x, g, b, None, 1e-3, A.ACT_RELU)
    by = x.numel() * 2
    tf = timeit(lambda: K.instnorm_fwd(x, g, b, None, 1e-3, A.ACT_RELU)) if DIR in ("both", "fwd") else float("nan")
    tb = timeit(lambda: K.instnorm_bwd(dy, x, g, b, st, dg, db, False, A.ACT_RELU)) if DIR in ("both", "bwd") else float("nan")
    print(f"{str(shape):22s} {by/1e6:7.1f} MB | fwd {tf:7.1f} us {3*by/tf/1e3:7.0f} GB/s | bwd {tb:7.1f} us {5*by/tb/1e3:7.0f} GB/s", flush=True)
