"""Per-call-site timing of every convolution of G and D at the bench shape (N=8, 256x512, bf16):
fwd / dgrad / wgrad microseconds and TFLOP/s (algorithmic, un-padded channel counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import labenv; labenv.select()
import sggan_amd
from sggan_amd import kernels as K

N, H, W = 8, 256, 512
dt = torch.bfloat16
# name, kind, Cin, Cout, R, stride, padding, reflect, Hin, Win, count per step (fwd, dgrad, wgrad)
L = [("G.c1 stem 7x7", "conv", 3, 64, 7, 1, "VALID", 3, H, W, (1, 0, 1)),
     ("G.c2 s2", "conv", 64, 128, 3, 2, "SAME", 0, H, W, (1, 1, 1)),
     ("G.c3 s2", "conv", 128, 256, 3, 2, "SAME", 0, H // 2, W // 2, (1, 1, 1)),
     ("G.res 3x3", "conv", 256, 256, 3, 1, "VALID", 1, H // 4, W // 4, (18, 18, 18)),
     ("G.d1 deconv", "deconv", 256, 128, 3, 2, None, 0, H // 4, W // 4, (1, 1, 1)),
     ("G.d2 deconv", "deconv", 128, 64, 3, 2, None, 0, H // 2, W // 2, (1, 1, 1)),
     ("G.out head 7x7", "conv", 64, 3, 7, 1, "VALID", 3, H, W, (1, 1, 1)),
     ("D.h0 s2", "conv", 3, 64, 3, 2, "SAME", 0, H, W, (2, 1, 2)),
     ("D.h1 s2", "conv", 64, 128, 3, 2, "SAME", 0, H // 2, W // 2, (2, 3, 2)),
     ("D.h2 s2", "conv", 128, 256, 3, 2, "SAME", 0, H // 4, W // 4, (2, 3, 2)),
     ("D.h3 s1", "conv", 256, 512, 3, 1, "SAME", 0, H // 8, W // 8, (2, 3, 2)),
     ("D.h31 s2v", "conv", 512, 512, 3, 2, "VALID", 0, 32, 64, (2, 3, 2)),
     ("D.h32 s2v", "conv", 512, 512, 3, 2, "VALID", 0, 15, 31, (2, 3, 2)),
     ("D.h33 s1v", "conv", 512, 512, 3, 1, "VALID", 0, 7, 15, (2, 3, 2)),
     ("D.h4 s1", "conv", 512, 34, 3, 1, "SAME", 0, 5, 13, (2, 3, 2))]


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print(f"{'layer':16s} {'GFLOP':>7s} | {'fwd us':>8s} {'TF/s':>6s} | {'dgrad us':>8s} {'TF/s':>6s} | {'wgrad us':>8s} {'TF/s':>6s} | per-step ms (f,d,w)")
for name, kind, ci, co, R, st, pad, refl, h, w, cnt in L:
    cip, cop = K.cpad(ci), K.cpad(co)
    if kind == "conv":
        g = K.conv_geom(N, h, w, cip, cop, R, R, st, pad, refl, dt)
        wt = torch.randn((R, R, ci, co), device="cuda") / (R * R * ci) ** 0.5
        wf, wd = K.pack_weights(wt, cip, cop, dt)
        x = torch.randn(g.x_shape, device="cuda").to(dt); dy = torch.randn(g.y_shape, device="cuda").to(dt)
        dw = torch.empty_like(wt)
        fl = 2.0 * g.y_shape[0] * g.y_shape[1] * g.y_shape[2] * co * R * R * ci
        fns = {"fwd": lambda: K.conv_fwd(g, x, wf, None), "dgrad": lambda: K.conv_dgrad(g, dy, wd), "wgrad": lambda: K.conv_wgrad(g, x, dy, dw)}
    else:
        g = K.deconv_geom(N, h, w, cip, cop, R, R, st, dt)
        wt = torch.randn((R, R, co, ci), device="cuda") / (R * R * ci) ** 0.5
        wf, wd = K.pack_weights(wt, cop, cip, dt)
        x = torch.randn(g.x_shape, device="cuda").to(dt); dy = torch.randn(g.y_shape, device="cuda").to(dt)
        dw = torch.empty_like(wt)
        fl = 2.0 * g.x_shape[0] * g.x_shape[1] * g.x_shape[2] * co * R * R * ci
        fns = {"fwd": lambda: K.deconv_fwd(g, x, wd, None), "dgrad": lambda: K.deconv_dgrad(g, dy, wf), "wgrad": lambda: K.deconv_wgrad(g, x, dy, dw)}
    row = f"{name:16s} {fl / 1e9:7.1f} |"
    ms = []
    for (op, c) in zip(("fwd", "dgrad", "wgrad"), cnt):
        us = timeit(fns[op])
        row += f" {us:8.1f} {fl / us / 1e6:6.0f} |"
        tot[op] += us * c / 1e3
        ms.append(us * c / 1e3)
    print(row + " " + " ".join(f"{m:6.3f}" for m in ms), flush=True)
print("per-step totals (ms):", {k: round(v, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()), 3))
