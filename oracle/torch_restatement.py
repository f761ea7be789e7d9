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


def generator_resnet(P, x, n_blocks=9, eps=1e-3):
    """module.py:219-269.  x NCHW."""
    def cin(n, h, **kw):
        return inorm(conv2d(h, P[n + "_w"], P[n + "_b"], **kw), P[n + "_g"], P[n + "_beta"], eps)
    h = F.relu(cin("c1", x, reflect=3))
    h = F.relu(cin("c2", h, stride=2, padding="SAME"))
    h = F.relu(cin("c3", h, stride=2, padding="SAME"))
    for i in range(1, n_blocks + 1):
        y = F.relu(cin(f"r{i}a", h, reflect=1))
        y = cin(f"r{i}b", y, reflect=1)
        h = y + h
    for n in ("d1", "d2"):
        h = F.relu(inorm(deconv2d(h, P[n + "_w"], P[n + "_b"]), P[n + "_g"], P[n + "_beta"], eps))
    return torch.tanh(conv2d(h, P["out_w"], P["out_b"], reflect=3))


def discriminator(P, x, mask, leak=0.3, eps=1e-3):
    """module.py:272-318.  x NCHW, mask NCHW (N,C,hm,wm)."""
    h = F.leaky_relu(conv2d(x, P["h0_w"], P["h0_b"], 2, "SAME"), leak)
    for n, s, p in (("h1", 2, "SAME"), ("h2", 2, "SAME"), ("h3", 1, "SAME"),
                    ("h31", 2, "VALID"), ("h32", 2, "VALID"), ("h33", 1, "VALID")):
        h = conv2d(h, P[n + "_w"], P[n + "_b"], s, p)
        h = F.leaky_relu(inorm(h, P[n + "_g"], P[n + "_beta"], eps), leak)
    h4 = conv2d(h, P["h4_w"], P["h4_b"], 1, "SAME")
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
        x, seg, mask = to(real_A), to(seg_A), to(mask_A)
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
