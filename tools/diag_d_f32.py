"""Per-layer accuracy of the f32 discriminator forward / backward against the float64 oracle (diagnostic):
where does the f32 parity path drift?   python tools/diag_d_f32.py [--dtype f32|bf16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sggan_amd
from sggan_amd import kernels as K
from oracle import sggan_oracle as O
from tests.golden.make_golden import d256_inputs

ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="f32"); a = ap.parse_args()
PG, PD, real, seg, mask = d256_inputs(23)
rng = np.random.default_rng(1)
dlog = rng.standard_normal((1, 5, 5, 1))
# oracle with every intermediate kept
t = O.Tape()
VD = {k: O.Var(v, k) for k, v in PD.items()}
x = O.Var(seg)
out = O.discriminator(t, VD, x, mask)
t.backward([(out, dlog)])
inter = [o for o, _ in t.ops]          # conv, lrelu, then (conv, IN, lrelu) x 6, conv(h4), mask_reduce

D = sggan_amd.Discriminator(df_dim=64, dtype=torch.float32 if a.dtype == "f32" else torch.bfloat16, device="cuda", seed=None)
D.P.load(PD)
rec = {"in_bwd": [], "dgrad": [], "in_fwd": []}
oi, od, of = K.instnorm_bwd, K.conv_dgrad, K.instnorm_fwd
def w_in(*args, **kw):
    r = oi(*args, **kw); rec["in_bwd"].append(r); return r
def w_dg(*args, **kw):
    r = od(*args, **kw); rec["dgrad"].append(r); return r
def w_if(*args, **kw):
    r = of(*args, **kw); rec["in_fwd"].append(r); return r
K.instnorm_bwd, K.conv_dgrad, K.instnorm_fwd = w_in, w_dg, w_if
xin = D.to_internal(torch.as_tensor(seg, dtype=torch.float32).cuda())
logits, tape = D.forward(xin, torch.as_tensor(mask, dtype=torch.float32).cuda())
D.P.zero_grad()
D.backward(tape, torch.as_tensor(dlog, dtype=torch.float32).cuda(), want_dx=True)
torch.cuda.synchronize()

def err(got, exp):
    got = got.detach().float().cpu().numpy().astype(np.float64)[..., :exp.shape[-1]]
    n = np.sqrt((exp ** 2).sum())
    return np.sqrt(((got - exp) ** 2).sum()) / max(n, 1e-30), np.abs(got - exp).max() / max(np.abs(exp).max(), 1e-30)

names = ["h0", "h1", "h2", "h3", "h31", "h32", "h33"]
print("forward: conv output (pre-norm) and unit output, relative L2 / max error")
oi_ = 0
for i, nme in enumerate(names):
    conv_o = inter[oi_]; oi_ += 1
    if i > 0:
        oi_ += 1                      # instance norm
    act_o = inter[oi_]; oi_ += 1
    g, xi, xc, stats = tape[i]
    e1 = err(xc, conv_o.v)
    nxt = tape[i + 1][1]
    e2 = err(nxt, act_o.v)
    extra = ""
    if stats is not None:
        mu = conv_o.v.mean((1, 2)); var = conv_o.v.var((1, 2)); rstd = 1 / np.sqrt(var + 1e-3)
        st = stats.cpu().numpy().astype(np.float64)[:, :mu.shape[-1]]
        extra = f"  mean err {np.abs(st[..., 0] - mu).max():.2e} (|mean| max {np.abs(mu).max():.2e}, std min {np.sqrt(var).min():.2e})  rstd rel err {np.abs(st[..., 1] / rstd - 1).max():.2e}"
    print(f"  {nme:4s} conv {e1[0]:.2e}/{e1[1]:.2e}   out {e2[0]:.2e}/{e2[1]:.2e}{extra}")
print("logits", err(logits, out.v))
print("backward: gradient wrt the conv output (after norm backward) and wrt the unit input; parameter gradients")
gD = D.P.export(D.P.grad)
k_in, k_dg = 0, 0
oi_ = len(inter) - 3        # h33's lrelu output index: inter[-1] mask_reduce, inter[-2] h4 conv, inter[-3] lrelu(h33)
# walk back: units in reverse
conv_idx = {}
j = 0
for i, nme in enumerate(names):
    conv_idx[nme] = j
    j += 2 if i == 0 else 3
dg_list = rec["dgrad"]           # order of calls: h4, h33, h32, ..., h0(if want_dx)
in_list = rec["in_bwd"]          # h33, h32, ..., h1
print("  h4 dx   ", err(dg_list[0], inter[conv_idx["h33"] + 2].g))
for r, nme in enumerate(reversed(names)):
    ci = conv_idx[nme]
    line = f"  {nme:4s}"
    if nme != "h0":
        line += f" d(conv out) {err(in_list[r], inter[ci].g)[0]:.2e}/{err(in_list[r], inter[ci].g)[1]:.2e}"
    src = x if nme == "h0" else inter[ci - 1]
    e = err(dg_list[1 + r], src.g)
    line += f"   d(input) {e[0]:.2e}/{e[1]:.2e}"
    for suf in ("_w", "_g", "_beta"):
        if nme + suf in gD and np.abs(VD[nme + suf].g).max() > 1e-12:
            gg = gD[nme + suf].astype(np.float64); ee = VD[nme + suf].g
            line += f"   {suf[1:]} {np.sqrt(((gg - ee) ** 2).sum()) / np.sqrt((ee ** 2).sum()):.2e}"
    print(line)

# where are the largest element errors of d(conv out), and is the pre-activation (norm output) ~0 there (LeakyReLU kink)?
for r, nme in enumerate(reversed(names)):
    if nme == "h0":
        continue
    ci = conv_idx[nme]
    got = in_list[r].detach().float().cpu().numpy().astype(np.float64)
    exp = inter[ci].g
    pre = inter[ci + 1].v                      # instance-norm output = LeakyReLU input
    gin = inter[ci + 2].g                      # gradient wrt the LeakyReLU output
    d = np.abs(got[..., :exp.shape[-1]] - exp)
    idx = np.argsort(d.ravel())[::-1][:3]
    print(nme, "top element errors:", [(float("%.2e" % d.ravel()[i]), "pre-act %.2e" % pre.ravel()[i], "g_in %.2e" % gin.ravel()[i]) for i in idx],
          " max|g| %.2e" % np.abs(exp).max(), " #|pre|<1e-6:", int((np.abs(pre) < 1e-6).sum()))
