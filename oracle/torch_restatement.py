"""Independent PyTorch-CPU composition of the reference-mode train step.

TEST INFRASTRUCTURE ONLY (see ``oracle/sggan_oracle.py`` header): imported by
``tests/`` as a second, independently-built statement of the same maths
(``F.pad`` + ``F.conv2d``, ``F.conv_transpose2d`` + crop, ``F.instance_norm``,
autograd) used to cross-check the NumPy oracle, and by ``bench.py``'s
``cpu_baseline`` leg as the "CPU restatement, not TF2" timing of
``model.py:169-200`` (the reference's own TF2 path cannot
run: tensorflow is not installed and there is no network).

PARITY UNPINNED for all floating-point results (no reference-side vectors).

Parameters use the reference's layouts (HWIO conv kernels, (kh,kw,out,in)
transpose kernels -- module.py:211-264) and are permuted to torch's on the fly.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _same_pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv2d(x, w, b, stride=1, padding="VALID", reflect=0):
    """x NCHW; w HWIO (module.py Conv2D call sites); TF SAME is asymmetric."""
    wt = w.permute(3, 2, 0, 1)
    if reflect:
        x = F.pad(x, (reflect,) * 4, mode="reflect")
    elif padding == "SAME":
        pt, pb = _same_pads(x.shape[2], w.shape[0], stride)
        pl, pr = _same_pads(x.shape[3], w.shape[1], stride)
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, wt, b, stride=stride)


def deconv2d(x, w, b, stride=2):
    """Conv2DTranspose SAME (module.py:254,258); w (kh,kw,out,in).  torch's
    conv_transpose2d with padding=0 yields the 'full' (H-1)*s+k map; TF keeps
    [before : before+s*H] with before = SAME-pad-before of the forward conv."""
    wt = w.permute(3, 2, 0, 1)  # (in, out, kh, kw)
    full = F.conv_transpose2d(x, wt, None, stride=stride)
    Ho, Wo = x.shape[2] * stride, x.shape[3] * stride
    pt, _ = _same_pads(Ho, w.shape[0], stride)
    pl, _ = _same_pads(Wo, w.shape[1], stride)
    return full[:, :, pt:pt + Ho, pl:pl + Wo] + b.view(1, -1, 1, 1)


def inorm(x, g, beta, eps=1e-3):
    """tfa InstanceNormalization; F.instance_norm refuses 1x1 maps (D's h33 at
    128x128 is 1x1, where the result is exactly beta), so spell it out."""
    var, mu = torch.var_mean(x, dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * g.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


# ---------------------------------------------------------------------------- storage emulation
# The MI355X performance path keeps every activation and every activation gradient it WRITES TO HBM in bfloat16 and computes in
# f32 in between (NHWC tensors in the storage dtype, f32 accumulation; packed GEMM weights in the storage
# dtype).  ``bf16_storage()`` makes this restatement do the same at exactly those points -- S() below marks each tensor the
# HIP path stores: conv outputs that feed an instance norm, layer outputs (after the norm / activation / residual add), the
# network input, and the GEMM's copy of the weights -- so that the timed bf16 step has a step-level oracle whose distance to
# it is NOT the bf16-vs-f32 distance of the forward function (tests/test_bf16_fidelity_cpu.py) but summation-order noise only.
class _RoundStore(torch.autograd.Function):
    """A tensor stored in bfloat16: the value is rounded on the way forward and its gradient -- stored in bfloat16 too -- on the
    way back."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundWeights(torch.autograd.Function):
    """The GEMM operand copy of an f32 master weight (sgg_pack_conv_weights): rounded value, gradient passed to the master as is
    (the weight gradient is accumulated and kept in f32)."""

    @staticmethod
    def forward(ctx, w):
        return w.to(torch.bfloat16).to(w.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


_STORE = {"on": False}


def S(t):
    """Storage point of the HIP path (identity unless bf16_storage() is active)."""
    return _RoundStore.apply(t) if _STORE["on"] else t


def W(w):
    return _RoundWeights.apply(w) if _STORE["on"] else w


class bf16_storage:
    """Context manager: evaluate the restatement with the MI355X bf16 path's storage points (see above)."""

    def __enter__(self):
        self._old = _STORE["on"]
        _STORE["on"] = True
        return self

    def __exit__(self, *exc):
        _STORE["on"] = self._old


def generator_resnet(P, x, n_blocks=9, eps=1e-3):
    """module.py:219-269.  x NCHW."""
    def cin(n, h, **kw):
        return inorm(S(conv2d(h, W(P[n + "_w"]), P[n + "_b"], **kw)), P[n + "_g"], P[n + "_beta"], eps)
    h = S(F.relu(cin("c1", x, reflect=3)))
    h = S(F.relu(cin("c2", h, stride=2, padding="SAME")))
    h = S(F.relu(cin("c3", h, stride=2, padding="SAME")))
    for i in range(1, n_blocks + 1):
        y = S(F.relu(cin(f"r{i}a", h, reflect=1)))
        y = cin(f"r{i}b", y, reflect=1)
        h = S(y + h)
    for n in ("d1", "d2"):
        h = S(F.relu(inorm(S(deconv2d(h, W(P[n + "_w"]), P[n + "_b"])), P[n + "_g"], P[n + "_beta"], eps)))
    return S(torch.tanh(conv2d(h, W(P["out_w"]), P["out_b"], reflect=3)))          # tanh is fused into the conv epilogue: one store


def discriminator(P, x, mask, leak=0.3, eps=1e-3):
    """module.py:272-318.  x NCHW, mask NCHW (N,C,hm,wm)."""
    h = S(F.leaky_relu(conv2d(x, W(P["h0_w"]), P["h0_b"], 2, "SAME"), leak))        # fused activation: one store
    for n, s, p in (("h1", 2, "SAME"), ("h2", 2, "SAME"), ("h3", 1, "SAME"),
                    ("h31", 2, "VALID"), ("h32", 2, "VALID"), ("h33", 1, "VALID")):
        h = S(conv2d(h, W(P[n + "_w"]), P[n + "_b"], s, p))
        h = S(F.leaky_relu(inorm(h, P[n + "_g"], P[n + "_beta"], eps), leak))
    h4 = S(conv2d(h, W(P["h4_w"]), P["h4_b"], 1, "SAME"))
    return (h4 * mask).sum(1, keepdim=True)


def adam_tf_(theta, g, m, v, t, lr=1e-3, b1=0.5, b2=0.999, eps=1e-7):
    """Keras Adam form (epsilon outside the bias correction), in place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    lr_t = lr * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
    theta.addcdiv_(m, v.sqrt().add_(eps), value=-lr_t)


class RefStep:
    """Stateful reference-mode step (model.py:169-200, deviations D2) on CPU."""

    def __init__(self, PG, PD, dtype=torch.float32, lr=1e-3, beta1=0.5, n_blocks=9):
        self.PG = {k: torch.as_tensor(v).to(dtype).clone().requires_grad_(True) for k, v in PG.items()}
        self.PD = {k: torch.as_tensor(v).to(dtype).clone().requires_grad_(True) for k, v in PD.items()}
        self.mG = {k: torch.zeros_like(v) for k, v in self.PG.items()}
        self.vG = {k: torch.zeros_like(v) for k, v in self.PG.items()}
        self.mD = {k: torch.zeros_like(v) for k, v in self.PD.items()}
        self.vD = {k: torch.zeros_like(v) for k, v in self.PD.items()}
        self.t, self.lr, self.beta1, self.n_blocks, self.dtype = 0, lr, beta1, n_blocks, dtype

    def step(self, real_A, seg_A, mask_A, apply=True):
        """NHWC numpy/tensor inputs (as model.py:250-256 feeds them)."""
        to = lambda a: torch.as_tensor(a).to(self.dtype).permute(0, 3, 1, 2).contiguous()
        x, seg, mask = S(to(real_A)), S(to(seg_A)), to(mask_A)          # images enter the networks in the storage dtype
        fake = generator_resnet(self.PG, x, self.n_blocks)
        da_real = discriminator(self.PD, seg, mask)
        da_fake = discriminator(self.PD, fake, mask)
        bce = F.binary_cross_entropy_with_logits
        gen_loss = bce(da_fake, torch.ones_like(da_fake)) + 100.0 * (seg - fake).abs().mean()
        disc_loss = bce(da_real, torch.ones_like(da_real)) + bce(da_fake, torch.zeros_like(da_fake))
        gG = torch.autograd.grad(gen_loss, list(self.PG.values()), retain_graph=True, allow_unused=True)
        gD = torch.autograd.grad(disc_loss, list(self.PD.values()), allow_unused=True)
        gG = [torch.zeros_like(p) if g is None else g for g, p in zip(gG, self.PG.values())]
        gD = [torch.zeros_like(p) if g is None else g for g, p in zip(gD, self.PD.values())]
        out = {"fake_A": fake.detach().permute(0, 2, 3, 1), "gen_loss": gen_loss.item(), "disc_loss": disc_loss.item(),
               "da_real": da_real.detach().permute(0, 2, 3, 1), "da_fake": da_fake.detach().permute(0, 2, 3, 1),
               "gG": dict(zip(self.PG, gG)), "gD": dict(zip(self.PD, gD))}
        if apply:
            self.t += 1
            with torch.no_grad():
                for (k, p), g in zip(self.PG.items(), gG):
                    adam_tf_(p, g, self.mG[k], self.vG[k], self.t, self.lr, self.beta1)
                for (k, p), g in zip(self.PD.items(), gD):
                    adam_tf_(p, g, self.mD[k], self.vD[k], self.t, self.lr, self.beta1)
        return out


# ---------------------------------------------------------------------------- cycle mode (2G+2D, deviation D5)
def seg_edge_weight(seg):
    """model.py:108-119 (NCHW): REFLECT-pad 1, central differences in x and y, |.| summed over channels, sign."""
    sp = F.pad(seg, (1, 1, 1, 1), mode="reflect")
    dx = sp[:, :, 1:-1, 2:] - sp[:, :, 1:-1, :-2]
    dy = sp[:, :, 2:, 1:-1] - sp[:, :, :-2, 1:-1]
    return (dx.abs() + dy.abs()).sum(1, keepdim=True).sign().abs()


def gradloss(x, target, weight):
    """gradloss_criterion / tf_deriv (module.py:325-351) in NCHW via a grouped (depthwise) conv, SAME zero padding."""
    C = x.shape[1]
    kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]], dtype=x.dtype)
    ky = torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]], dtype=x.dtype)
    k = torch.stack([kx, ky]).repeat(C, 1, 1).unsqueeze(1)            # (2C,1,3,3): channel c -> outputs 2c, 2c+1
    d = lambda t: F.conv2d(t, k, padding=1, groups=C)
    a = (d(x).abs() - d(target).abs()).abs().mean(1, keepdim=True)
    return (weight * a).mean()


class CycleStep:
    """Stateful cycle-mode step on CPU -- same definition as oracle.sggan_oracle.cycle_step."""

    def __init__(self, P, dtype=torch.float32, lr=2e-4, beta1=0.5, L1_lambda=10.0, Lg_lambda=5.0, use_lsgan=True, n_blocks=9):
        mk = lambda d: {k: torch.as_tensor(v).to(dtype).clone().requires_grad_(True) for k, v in d.items()}
        self.P = {n: mk(P[n]) for n in ("Gab", "Gba", "Da", "Db")}
        self.m = {n: {k: torch.zeros_like(v) for k, v in p.items()} for n, p in self.P.items()}
        self.v = {n: {k: torch.zeros_like(v) for k, v in p.items()} for n, p in self.P.items()}
        self.t, self.lr, self.beta1, self.dtype, self.n_blocks = 0, lr, beta1, dtype, n_blocks
        self.L1, self.Lg, self.use_lsgan = L1_lambda, Lg_lambda, use_lsgan

    def step(self, real_A, real_B, seg_A, seg_B, mask_A, mask_B):
        to = lambda a: torch.as_tensor(a).to(self.dtype).permute(0, 3, 1, 2).contiguous()
        rA, rB, sA, sB, mA, mB = map(to, (real_A, real_B, seg_A, seg_B, mask_A, mask_B))
        rA, rB, sA, sB = S(rA), S(rB), S(sA), S(sB)                      # images enter the networks in the storage dtype
        P, nb = self.P, self.n_blocks
        fake_B = generator_resnet(P["Gab"], rA, nb); cyc_A = generator_resnet(P["Gba"], fake_B, nb)
        fake_A = generator_resnet(P["Gba"], rB, nb); cyc_B = generator_resnet(P["Gab"], fake_A, nb)
        DB_fake = discriminator(P["Db"], fake_B, mA); DA_fake = discriminator(P["Da"], fake_A, mB)
        DA_real = discriminator(P["Da"], rA, mA); DB_real = discriminator(P["Db"], rB, mB)
        if self.use_lsgan:
            crit = lambda x, z: ((x - z) ** 2).mean()
        else:
            crit = lambda x, z: F.binary_cross_entropy_with_logits(x, torch.full_like(x, z))
        wA, wB = seg_edge_weight(sA), seg_edge_weight(sB)
        g_loss = (crit(DA_fake, 1.0) + crit(DB_fake, 1.0) + self.L1 * ((rA - cyc_A).abs().mean() + (rB - cyc_B).abs().mean())
                  + self.Lg * (gradloss(fake_A, rB, wB) + gradloss(fake_B, rA, wA)))
        d_loss = 0.5 * (crit(DA_real, 1.0) + crit(DA_fake, 0.0)) + 0.5 * (crit(DB_real, 1.0) + crit(DB_fake, 0.0))
        gp = list(P["Gab"].values()) + list(P["Gba"].values())
        dp = list(P["Da"].values()) + list(P["Db"].values())
        gg = torch.autograd.grad(g_loss, gp, retain_graph=True, allow_unused=True)
        gd = torch.autograd.grad(d_loss, dp, allow_unused=True)
        grads = {}
        it = iter([torch.zeros_like(p) if g is None else g for g, p in zip(list(gg) + list(gd), gp + dp)])
        for n in ("Gab", "Gba", "Da", "Db"):
            grads[n] = {k: next(it) for k in P[n]}
        self.t += 1
        with torch.no_grad():
            for n in P:
                for k, p in P[n].items():
                    adam_tf_(p, grads[n][k], self.m[n][k], self.v[n][k], self.t, self.lr, self.beta1)
        return {"g_loss": g_loss.item(), "d_loss": d_loss.item(), "grads": grads,
                "fake_B": fake_B.detach().permute(0, 2, 3, 1), "cyc_A": cyc_A.detach().permute(0, 2, 3, 1)}
