"""D pair: one stacked pass over [real_B; fake_B | fake_A; real_A] vs two passes; where do the backward signals start to differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sggan_amd import kernels as K
from sggan_amd.module import Discriminator, DiscriminatorPair
dt = torch.float32 if sys.argv[1] == "f32" else torch.bfloat16
n, H, W = int(sys.argv[2]), 256, 512
Da = Discriminator(df_dim=64, dtype=dt, seed=21); Db = Discriminator(df_dim=64, dtype=dt, seed=23)
gen = torch.Generator().manual_seed(5)
img = lambda: K.pad_channels((torch.rand((n, H, W, 3), generator=gen) * 2 - 1).cuda(), 8, dt)
rA, rB, fA, fB = img(), img(), img(), img()
mh, mw = Da.out_hw(H, W)
msk = lambda: (torch.rand((n, mh, mw, 34), generator=gen) > 0.5).float().cuda()
mA, mB = msk(), msk()
gl = lambda t: torch.randn(t.shape, generator=gen).cuda()

def run(quad):
    for D in (Da, Db):
        D.P.zero_grad()
    P = DiscriminatorPair(Db, Da)
    sig = {}
    if quad:
        out, tape = P.forward(torch.cat([rB, fB, fA, rA]), torch.cat([mB, mA, mB, mA]))
        g = G4
        d = K.mask_reduce_bwd(g.contiguous(), tape[-1][0], tape[-1][1], dt, 34)
        units = P.units + [P.h4]
        for i in range(len(units) - 1, -1, -1):
            d = units[i].backward(tape[i], d, True, True)
            sig[i] = d
    else:
        P2 = DiscriminatorPair(Da, Db)
        outr, tr = P2.forward(torch.cat([rA, rB]), torch.cat([mA, mB]))      # [D_A(rA); D_B(rB)]
        outf, tf = P.forward(torch.cat([fB, fA]), torch.cat([mA, mB]))       # [D_B(fB); D_A(fA)]
        g_r = torch.cat([G4[3 * n:], G4[:n]]); g_f = G4[n:3 * n]
        for PP, tp, g, tag in ((P2, tr, g_r, "r"), (P, tf, g_f, "f")):
            d = K.mask_reduce_bwd(g.contiguous(), tp[-1][0], tp[-1][1], dt, 34)
            units = PP.units + [PP.h4]
            for i in range(len(units) - 1, -1, -1):
                d = units[i].backward(tp[i], d, True, True)
                sig[(tag, i)] = d
        out = None
    return sig, {f"{nm}.{k}": D.P.g(k).clone() for nm, D in (("Da", Da), ("Db", Db)) for k in D.P.names()}

o, _ = DiscriminatorPair(Db, Da).forward(torch.cat([rB, fB, fA, rA]), torch.cat([mB, mA, mB, mA]))
G4 = gl(o)
s0, g0 = run(False)
s1, g1 = run(True)
for k in g0:
    rel = float((g0[k].double() - g1[k].double()).norm() / (g0[k].double().norm() + 1e-30))
    if rel > 1e-5: print(f"grad {k:14s} rel {rel:.2e}")
for i in sorted(s1):
    q = s1[i]                       # [rB; fB | fA; rA]
    ref = torch.cat([s0[("r", i)][n:], s0[("f", i)], s0[("r", i)][:n]])
    per = [float((q[j * n:(j + 1) * n].double() - ref[j * n:(j + 1) * n].double()).norm() / (ref[j * n:(j + 1) * n].double().norm() + 1e-30)) for j in range(4)]
    print("signal after unit", i, ["%.1e" % p for p in per])
