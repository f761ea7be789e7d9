"""Per-tensor difference of the parameter gradients of the paired cycle step with and without the stacked real + fake discriminator pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sggan_amd as sg
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
dtype, width, blocks, N, H, W = sys.argv[1], int(sys.argv[2]), 2, int(sys.argv[3]), 256, 512

def rand_inputs(N, H, W, D, seed):
    g = torch.Generator().manual_seed(seed)
    mh, mw = D.out_hw(H, W)
    return (torch.rand((N, H, W, 3), generator=g) * 2 - 1, torch.rand((N, H, W, 3), generator=g) * 2 - 1,
            (torch.rand((N, mh, mw, 34), generator=g) > 0.5).float())

res = []
for quad in (False, True):
    m = sg.sggan(sg.default_args(ngf=width, ndf=width, n_blocks=blocks, dtype=dtype, cycle=True, paired=True, d_quad=quad))
    m.real_A, m.seg_A, m.mask_A = rand_inputs(N, H, W, m.discriminator, 61)
    m.real_B, m.seg_B, m.mask_B = rand_inputs(N, H, W, m.discriminator, 62)
    m.train_step()
    d = {}
    for k, net in enumerate(m.networks()):
        for name in net.P.names():
            d[f"net{k}.{name}"] = net.P.g(name).clone()
    res.append(d)
for k in res[0]:
    a, b = res[0][k].double(), res[1][k].double()
    rel = float((a - b).norm() / (a.norm() + 1e-30))
    if rel > 1e-5:
        print(f"{k:24s} rel {rel:.3e}  |a| {float(a.norm()):.3e}")
print("done")
