"""The reference's primitive vocabulary -- ``conv2d / deconv2d / instance_norm / lrelu`` (ops.py:13-37;
conv2d/deconv2d are commented out there at :24-34, the live layers are the Keras ones in module.py) --
as differentiable ops over the HIP kernels.  Numerics follow the LIVE path by default
(InstanceNorm eps=1e-3, LeakyReLU alpha=0.3); pass ``eps=1e-5`` / ``leak=0.2`` for the ops.py spec.

Tensors are NHWC on the GPU, float32 or bfloat16.  Channel counts that are not a multiple of 8 are
zero-padded on the way in and cropped on the way out (the kernels work on 16-byte channel chunks).
Weights use the reference layouts: HWIO for conv2d, (kh,kw,out,in) for deconv2d; parameters and their
gradients are float32.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _abi as A
from . import kernels as K


def _padc(x, cp):
    return x if x.shape[-1] == cp else F.pad(x, (0, cp - x.shape[-1]))


def _vec(v, cp):
    v = v.to(torch.float32)
    return (v if v.numel() == cp else F.pad(v, (0, cp - v.numel()))).contiguous()


def _parse_padding(padding):
    """'SAME' | 'VALID' | 'REFLECT-p' (tf.pad REFLECT of p, then VALID)."""
    if isinstance(padding, str) and padding.upper().startswith("REFLECT"):
        return "VALID", int(padding.split("-")[1])
    return padding.upper(), 0


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, reflect, act, leak):
        R, S, Cr, Kr = w.shape
        Cp, Kp = K.cpad(Cr), K.cpad(Kr)
        xp = _padc(x, Cp).contiguous()
        N, H, W, _ = xp.shape
        g = K.conv_geom(N, H, W, Cp, Kp, R, S, stride, padding, reflect, xp.dtype)
        wf, wd = K.pack_weights(w.detach().to(torch.float32).contiguous(), Cp, Kp, xp.dtype)
        y = K.conv_fwd(g, xp, wf, None if b is None else _vec(b.detach(), Kp), act, leak)
        ctx.g, ctx.act, ctx.leak, ctx.dims, ctx.has_b = g, act, leak, (Cr, Kr, x.shape[-1]), b is not None
        ctx.save_for_backward(xp, wd, y)
        return y[..., :Kr] if Kp != Kr else y

    @staticmethod
    def backward(ctx, dy):
        xp, wd, y = ctx.saved_tensors
        Cr, Kr, Cx = ctx.dims
        g = ctx.g
        dy = _padc(dy.to(xp.dtype), g.y_shape[-1]).contiguous()
        if ctx.act != A.ACT_NONE:
            dy = K.act_bwd(dy, y, ctx.act, ctx.leak)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = K.conv_dgrad(g, dy, wd)[..., :Cx]
        if ctx.needs_input_grad[1]:
            dw = torch.empty((g.desc.R, g.desc.S, Cr, Kr), dtype=torch.float32, device=dy.device)
            K.conv_wgrad(g, xp, dy, dw)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = torch.empty(Kr, dtype=torch.float32, device=dy.device)
            K.bias_grad(dy, db)
        return dx, dw, db, None, None, None, None, None


class _Deconv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, act, leak):
        R, S, Co, Ci = w.shape
        Cop, Cip = K.cpad(Co), K.cpad(Ci)
        xp = _padc(x, Cip).contiguous()
        N, H, W, _ = xp.shape
        g = K.deconv_geom(N, H, W, Cip, Cop, R, S, stride, xp.dtype)
        wf, wd = K.pack_weights(w.detach().to(torch.float32).contiguous(), Cop, Cip, xp.dtype)
        y = K.deconv_fwd(g, xp, wd, None if b is None else _vec(b.detach(), Cop), act, leak)
        ctx.g, ctx.act, ctx.leak, ctx.dims, ctx.has_b = g, act, leak, (Co, Ci, x.shape[-1]), b is not None
        ctx.save_for_backward(xp, wf, y)
        return y[..., :Co] if Cop != Co else y

    @staticmethod
    def backward(ctx, dy):
        xp, wf, y = ctx.saved_tensors
        Co, Ci, Cx = ctx.dims
        g = ctx.g
        dy = _padc(dy.to(xp.dtype), g.y_shape[-1]).contiguous()
        if ctx.act != A.ACT_NONE:
            dy = K.act_bwd(dy, y, ctx.act, ctx.leak)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = K.deconv_dgrad(g, dy, wf)[..., :Cx]
        if ctx.needs_input_grad[1]:
            dw = torch.empty((g.desc.R, g.desc.S, Co, Ci), dtype=torch.float32, device=dy.device)
            K.deconv_wgrad(g, xp, dy, dw)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = torch.empty(Co, dtype=torch.float32, device=dy.device)
            K.bias_grad(dy, db)
        return dx, dw, db, None, None, None


class _InstanceNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, leak):
        Cr = x.shape[-1]
        Cp = K.cpad(Cr)
        xp = _padc(x, Cp).contiguous()
        g, b = _vec(gamma.detach(), Cp), _vec(beta.detach(), Cp)
        y, stats = K.instnorm_fwd(xp, g, b, None, eps, act, leak)
        ctx.act, ctx.leak, ctx.Cr = act, leak, Cr
        ctx.save_for_backward(xp, g, b, stats)
        return y[..., :Cr] if Cp != Cr else y

    @staticmethod
    def backward(ctx, dy):
        xp, g, b, stats = ctx.saved_tensors
        dy = _padc(dy.to(xp.dtype), xp.shape[-1]).contiguous()
        dg = torch.empty(ctx.Cr, dtype=torch.float32, device=dy.device)
        db = torch.empty(ctx.Cr, dtype=torch.float32, device=dy.device)
        dx = K.instnorm_bwd(dy, xp, g, b, stats, dg, db, False, ctx.act, ctx.leak)
        return dx[..., :ctx.Cr], dg, db, None, None, None


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, leak):
        Cr = x.shape[-1]
        xp = _padc(x, K.cpad(Cr)).contiguous()
        y = K.act_fwd(xp, act, leak)
        ctx.act, ctx.leak, ctx.Cr = act, leak, Cr
        ctx.save_for_backward(y)
        return y[..., :Cr] if y.shape[-1] != Cr else y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _padc(dy.to(y.dtype), y.shape[-1]).contiguous()
        return K.act_bwd(dy, y, ctx.act, ctx.leak)[..., :ctx.Cr], None, None


_ACTS = {None: A.ACT_NONE, "none": A.ACT_NONE, "relu": A.ACT_RELU, "lrelu": A.ACT_LRELU, "tanh": A.ACT_TANH}


def conv2d(x, w, b=None, stride=1, padding="SAME", act=None, leak=0.3):
    """tf.keras.layers.Conv2D(k, (R,S), strides, padding) [+ tf.pad REFLECT via padding='REFLECT-p'] (module.py:210-311).
    TF 'SAME' is asymmetric (extra pad bottom/right)."""
    pad, reflect = _parse_padding(padding)
    return _Conv2dFn.apply(x, w, b, int(stride), pad, reflect, _ACTS[act], float(leak))


def deconv2d(x, w, b=None, stride=2, act=None, leak=0.3):
    """tf.keras.layers.Conv2DTranspose(k, (3,3), strides=(2,2), padding='same') (module.py:254,258); w is (kh,kw,out,in)."""
    return _Deconv2dFn.apply(x, w, b, int(stride), _ACTS[act], float(leak))


def instance_norm(x, gamma, beta, eps=1e-3, act=None, leak=0.3):
    """tfa.layers.InstanceNormalization (module.py:212..308); eps=1e-5 gives ops.instance_norm (ops.py:13-22)."""
    return _InstanceNormFn.apply(x, gamma, beta, float(eps), _ACTS[act], float(leak))


def lrelu(x, leak=0.3):
    """tf.keras.layers.LeakyReLU() alpha=0.3 (module.py:285-309); leak=0.2 gives ops.lrelu (ops.py:36-37)."""
    return _ActFn.apply(x, A.ACT_LRELU, float(leak))


def relu(x):
    return _ActFn.apply(x, A.ACT_RELU, 0.0)


def tanh(x):
    return _ActFn.apply(x, A.ACT_TANH, 0.0)
