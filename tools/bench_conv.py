"""Micro-benchmark of one convolution call site (fwd / dgrad / wgrad) through the C ABI.
    python tools/bench_conv.py [--n 8 --h 64 --w 128 --c 256 --k 256 --r 3 --stride 1 --pad REFLECT-1 --iters 50 --ops fwd,dgrad,wgrad]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import labenv; labenv.select()
import sggan_amd
from sggan_amd import kernels as K

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8); ap.add_argument("--h", type=int, default=64); ap.add_argument("--w", type=int, default=128)
ap.add_argument("--c", type=int, default=256); ap.add_argument("--k", type=int, default=256); ap.add_argument("--r", type=int, default=3)
ap.add_argument("--stride", type=int, default=1); ap.add_argument("--pad", default="REFLECT-1")
ap.add_argument("--iters", type=int, default=50); ap.add_argument("--ops", default="fwd,dgrad,wgrad"); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--rounds", type=int, default=1)
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
pad, refl = ("VALID", int(a.pad.split("-")[1])) if a.pad.startswith("REFLECT") else (a.pad, 0)
Cp, Kp = K.cpad(a.c), K.cpad(a.k)          # channel counts are stored padded to 8 (3 -> 8)
g = K.conv_geom(a.n, a.h, a.w, Cp, Kp, a.r, a.r, a.stride, pad, refl, dt)
x = torch.randn(g.x_shape, device="cuda").to(dt)
dy = torch.randn(g.y_shape, device="cuda").to(dt)
w = torch.randn((a.r, a.r, a.c, a.k), device="cuda") / (a.r * a.r * a.c) ** 0.5
wf, wd = K.pack_weights(w, Cp, Kp, dt)
dw = torch.empty_like(w)
flops = 2.0 * g.y_shape[0] * g.y_shape[1] * g.y_shape[2] * a.k * a.r * a.r * a.c
fns = {"fwd": lambda: K.conv_fwd(g, x, wf, None), "dgrad": lambda: K.conv_dgrad(g, dy, wd), "wgrad": lambda: K.conv_wgrad(g, x, dy, dw),
       "dgrad_add": lambda: K.conv_dgrad(g, dy, wd, x)}
if "dgrad_stats" in a.ops or "in_partial" in a.ops:      # data gradient + the norm-backward sums of the norm in front of it, vs the separate pass
    from sggan_amd import _abi as AB
    gam, bet = torch.ones(Cp, device="cuda"), torch.zeros(Cp, device="cuda")
    nx = torch.randn(g.x_shape, device="cuda").to(dt)
    _, nst = K.instnorm_fwd(nx, gam, bet, None, 1e-3, AB.ACT_RELU)
    dgm, dbt = torch.empty(Cp, device="cuda"), torch.empty(Cp, device="cuda")
    fns["dgrad_stats"] = lambda: K.conv_dgrad_stats(g, dy, wd, x, nx, nst, gam, bet, AB.ACT_RELU, 0.0)
    dxs, part = K.conv_dgrad_stats(g, dy, wd, x, nx, nst, gam, bet, AB.ACT_RELU, 0.0)
    fns["in_bwd"] = lambda: K.instnorm_bwd(dxs, nx, gam, bet, nst, dgm, dbt, False, AB.ACT_RELU)
    fns["in_bwd_partial"] = lambda: K.instnorm_bwd_partial(dxs, nx, part, gam, bet, nst, dgm, dbt, False, AB.ACT_RELU)
if any(o in a.ops.split(",") for o in ("fwd_pair", "dgrad_pair", "wgrad_pair2", "fwd_normload_pair", "in_apply_pair", "fwd_stats", "fwd_normload", "in_apply")):
    # the launches the paired cycle step makes: a stacked batch of two networks (images [:n/2] / [n/2:]) with two weight sets
    assert a.n % 2 == 0
    h = a.n // 2
    w2 = torch.randn((a.r, a.r, a.c, a.k), device="cuda") / (a.r * a.r * a.c) ** 0.5
    wf2, wd2 = K.pack_weights(w2, Cp, Kp, dt)
    bias = torch.zeros(Kp, device="cuda")
    fns["fwd_pair"] = lambda: K.conv_fwd_stats_pair(g, x, wf, bias, wf2, bias, h)
    fns["dgrad_pair"] = lambda: K.conv_dgrad_pair(g, dy, wd, wd2, h, x)                      # + the skip-gradient addend, as in the step
    # normalise-on-load: [finalize + apply pass] + conv  vs  finalize + the conv that applies the norm to its operand tiles
    from sggan_amd import _abi as AB
    gam, bet = torch.rand(Cp, device="cuda") + 0.5, torch.randn(Cp, device="cuda") * 0.1
    xf = x.float().reshape(a.n, 4, a.h * a.w // 4, Cp)
    part = torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=-1).contiguous()
    nstats = K.instnorm_finalize(part, a.h * a.w, 1e-3)
    fns["in_apply_pair"] = lambda: K.instnorm_fwd_partial_pair(x, part, gam, bet, gam, bet, h, None, 1e-3, AB.ACT_RELU, 0.0)
    fns["in_apply"] = lambda: K.instnorm_fwd_partial(x, part, gam, bet, None, 1e-3, AB.ACT_RELU, 0.0)
    fns["in_finalize"] = lambda: K.instnorm_finalize(part, a.h * a.w, 1e-3)
    fns["fwd_stats"] = lambda: K.conv_fwd_stats(g, x, wf, bias)
    fns["fwd_normload"] = lambda: K.conv_fwd_stats_normload(g, x, nstats, gam, bet, wf, bias)
    fns["fwd_normload_pair"] = lambda: K.conv_fwd_stats_normload(g, x, nstats, gam, bet, wf, bias, pair=(gam, bet, wf2, bias, h))
    gh = K.conv_geom(h, a.h, a.w, Cp, Kp, a.r, a.r, a.stride, pad, refl, dt)
    xs = [torch.randn(gh.x_shape, device="cuda").to(dt) for _ in range(4)]
    ds = [torch.randn(gh.y_shape, device="cuda").to(dt) for _ in range(4)]
    dw2 = torch.empty_like(w)
    fns["wgrad_pair2"] = lambda: K.conv_wgrad_pair2(gh, (xs[0], ds[0], xs[1], ds[1], dw), (xs[2], ds[2], xs[3], ds[3], dw2))
    FLOPS = {"wgrad_pair2": 4 * 2.0 * gh.y_shape[0] * gh.y_shape[1] * gh.y_shape[2] * a.k * a.r * a.r * a.c}
else:
    FLOPS = {}
def timed(f, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


ops = a.ops.split(",")
for op in ops:
    for _ in range(5):
        fns[op]()
torch.cuda.synchronize()
# --rounds R: the ops are timed round-robin R times and the median per op is printed (the clocks move with what ran before:
# an op timed first in a fresh process and the same op timed after others differ by up to 10 %)
res = {op: [] for op in ops}
for _ in range(a.rounds):
    for op in ops:
        res[op].append(timed(fns[op], a.iters))
for op in ops:
    ms = sorted(res[op])[len(res[op]) // 2]
    fl = FLOPS.get(op, flops)
    print(f"{op:18s} {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TFLOP/s  ({fl/1e9:.1f} GFLOP)  [min {min(res[op])*1e3:.1f}, max {max(res[op])*1e3:.1f}, {a.rounds} x {a.iters}]", flush=True)
