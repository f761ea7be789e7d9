"""Is a conv data gradient (and forward / weight gradient) per image independent of the batch size?  (h31-like 4x4 stride-2 VALID)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sggan_amd import kernels as K
for dt in (torch.float32, torch.bfloat16):
    for (H, W, Ci, Co, R, st, pad) in ((32, 64, 256, 512, 4, 2, "VALID"), (15, 31, 512, 512, 4, 2, "VALID"), (6, 14, 512, 512, 3, 1, "VALID"), (32, 64, 256, 512, 3, 1, "SAME")):
        gen = torch.Generator().manual_seed(3)
        w = torch.randn((R, R, Ci, Co), generator=gen).cuda() / (R * R * Ci) ** 0.5
        wf, wd = K.pack_weights(w, Ci, Co, dt)
        g1 = K.conv_geom(1, H, W, Ci, Co, R, R, st, pad, 0, dt)
        x1 = torch.randn(g1.x_shape, generator=gen).cuda().to(dt)
        dy1 = torch.randn(g1.y_shape, generator=gen).cuda().to(dt)
        ref_y, ref_dx = K.conv_fwd(g1, x1, wf, None), K.conv_dgrad(g1, dy1, wd)
        for N in (2, 4):
            g = K.conv_geom(N, H, W, Ci, Co, R, R, st, pad, 0, dt)
            y = K.conv_fwd(g, x1.repeat(N, 1, 1, 1), wf, None)
            dx = K.conv_dgrad(g, dy1.repeat(N, 1, 1, 1), wd)
            ey = max(float((y[i].float() - ref_y[0].float()).norm() / ref_y.float().norm()) for i in range(N))
            ed = max(float((dx[i].float() - ref_dx[0].float()).norm() / ref_dx.float().norm()) for i in range(N))
            # grouped: two "networks" with the same weights over 2N images
            y2 = K.conv_fwd_group2(g, x1.repeat(2 * N, 1, 1, 1), wf, None, wf, None, 0, 0.0)
            dx2 = K.conv_dgrad_group2(g, dy1.repeat(2 * N, 1, 1, 1), wd, wd, None)
            ey2 = max(float((y2[i].float() - ref_y[0].float()).norm() / ref_y.float().norm()) for i in range(2 * N))
            ed2 = max(float((dx2[i].float() - ref_dx[0].float()).norm() / ref_dx.float().norm()) for i in range(2 * N))
            print(f"{str(dt):15s} {H}x{W} {Ci}->{Co} {R}x{R} s{st} {pad:5s} N={N}: fwd {ey:.2e} dgrad {ed:.2e} | group2 fwd {ey2:.2e} dgrad {ed2:.2e}", flush=True)
