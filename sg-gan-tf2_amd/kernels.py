"""Tensor-level launchers over the C ABI (one function per entry point of include/sggan.h).

torch is used for device memory and streams only: every function passes raw device
pointers + the current HIP stream to libsggan.so.  Tensors are NHWC, channel-padded
to a multiple of 8 (see include/sggan.h), float32 (parity path) or bfloat16 (perf path).
"""
from __future__ import annotations

import ctypes as C
import functools

import torch

from . import _abi as A

_DT = {torch.float32: A.SGG_F32, torch.bfloat16: A.SGG_BF16}


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    try:
        return _DT[d]
    except KeyError:
        raise TypeError(f"unsupported dtype {d}: activations must be float32 or bfloat16")


def cpad(c: int) -> int:
    return (c + A.CPAD - 1) // A.CPAD * A.CPAD


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _s():
    """The current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() builds a Stream
    object per call (~8 us, 750 launches per step); the raw accessor is one C call."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_WS = {}

# Optional timing hook (bench.py): PROFILE(op_name, geom_or_shape) -> object with start()/stop(), or None.
# Events are recorded on the current stream, i.e. the stream the kernel is launched on.
PROFILE = None


def _prof(name, key):
    return PROFILE(name, key) if PROFILE is not None else None


# Host-side actions in step order (graph.StepProgram.host): while a step is being recorded into HIP graphs the recorder is
# installed here; an action then ends the current graph segment and is replayed between segments.  Otherwise it just runs.
_RECORDER = None


def host(fn):
    return fn() if _RECORDER is None else _RECORDER.host(fn)



def workspace(nbytes: int, device) -> torch.Tensor:
    """Per-device scratch buffer shared by all launches (single stream => in-order reuse is safe)."""
    key = torch.device(device).index or 0
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def workspace_refs():
    """The scratch buffers currently in use (a recorded StepProgram holds them: their addresses are in its launches)."""
    return list(_WS.values())


def same_pads(n_in: int, k: int, s: int):
    """TF 'SAME' (leading pad, trailing pad, output size): the extra pad goes bottom/right."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2, total - total // 2, out


class ConvGeom:
    """Geometry of one Conv2D / Conv2DTranspose call site, resolved to an sgg_conv_desc."""
    __slots__ = ("desc", "x_shape", "y_shape", "ws_wgrad", "ws_dgrad", "ws_fwd", "dtype", "is_deconv", "stats_chunks", "wgrad_pair", "bwd_stats_chunks", "dgrad_mixed", "pair_ok", "normload_ok")

    def __init__(self, desc, x_shape, y_shape, dtype, is_deconv):
        self.desc, self.x_shape, self.y_shape, self.dtype, self.is_deconv = desc, x_shape, y_shape, dtype, is_deconv
        self.ws_wgrad = int(A.lib().sgg_conv2d_bwd_weight_workspace(C.byref(desc)))
        L = A.lib()
        # pixel chunks of the (sum, sumsq) rows the forward conv can emit for a following instance norm (0: it cannot)
        self.stats_chunks = int((L.sgg_deconv2d_fwd_stats_chunks if is_deconv else L.sgg_conv2d_fwd_stats_chunks)(C.byref(desc)))
        # the weight gradients of two applications of the layer can share one launch (cycle step)
        self.wgrad_pair = (not is_deconv) and bool(L.sgg_conv2d_bwd_weight_pair_supported(C.byref(desc)))
        # chunks of the norm-backward partial sums the data gradient can emit for the norm that consumes dx (0: it cannot)
        self.bwd_stats_chunks = 0 if is_deconv else int(L.sgg_conv2d_bwd_data_stats_chunks(C.byref(desc)))
        # mixed mode: the data gradient can be written in f32 (bf16 operands) for the norm backward that consumes it
        self.dgrad_mixed = (not is_deconv) and bool(L.sgg_conv2d_bwd_data_mixed_supported(C.byref(desc)))
        # a stacked batch of two networks can run forward (+stats) / data gradient as ONE launch with per-image weights
        self.pair_ok = (not is_deconv) and bool(L.sgg_conv2d_pair_supported(C.byref(desc)))
        self.normload_ok = (not is_deconv) and bool(L.sgg_conv2d_fwd_normload_supported(C.byref(desc)))
        # workspaces of the two GEMM directions; a deconv runs the conv's data-gradient kernel forwards
        self.ws_fwd = int((L.sgg_deconv2d_fwd_workspace if is_deconv else L.sgg_conv2d_fwd_workspace)(C.byref(desc)))
        self.ws_dgrad = int((L.sgg_deconv2d_bwd_data_workspace if is_deconv else L.sgg_conv2d_bwd_data_workspace)(C.byref(desc)))
        if self.ws_wgrad == 0:
            raise A.SggError(f"invalid convolution geometry {[(f, getattr(desc, f)) for f, _ in desc._fields_]}")


@functools.lru_cache(maxsize=None)
def conv_geom(N, H, W, C_, K, R, S, stride, padding, reflect, dtype) -> ConvGeom:
    """tf.keras.layers.Conv2D geometry: padding 'SAME'|'VALID', or reflect=p (tf.pad REFLECT then VALID)."""
    if reflect:
        assert padding == "VALID" and stride == 1
        pt = pl = reflect
        Ho, Wo = H + 2 * reflect - R + 1, W + 2 * reflect - S + 1
        mode = A.PAD_REFLECT
    elif padding == "SAME":
        pt, _, Ho = same_pads(H, R, stride)
        pl, _, Wo = same_pads(W, S, stride)
        mode = A.PAD_ZERO
    elif padding == "VALID":
        pt = pl = 0
        Ho, Wo = (H - R) // stride + 1, (W - S) // stride + 1
        mode = A.PAD_ZERO
    else:
        raise ValueError(padding)
    if Ho <= 0 or Wo <= 0:
        raise ValueError(f"convolution output is empty for input {H}x{W}, kernel {R}x{S}, stride {stride}, {padding}")
    d = A.ConvDesc(N, H, W, C_, K, R, S, stride, pt, pl, Ho, Wo, mode, dt(dtype))
    return ConvGeom(d, (N, H, W, C_), (N, Ho, Wo, K), dtype, False)


@functools.lru_cache(maxsize=None)
def deconv_geom(N, Hin, Win, Cin, Cout, R, S, stride, dtype) -> ConvGeom:
    """Conv2DTranspose(padding='same'): described by the equivalent conv whose INPUT is the deconv OUTPUT."""
    H, W = Hin * stride, Win * stride
    pt, _, oh = same_pads(H, R, stride)
    pl, _, ow = same_pads(W, S, stride)
    assert oh == Hin and ow == Win
    d = A.ConvDesc(N, H, W, Cout, Cin, R, S, stride, pt, pl, Hin, Win, A.PAD_ZERO, dt(dtype))
    return ConvGeom(d, (N, Hin, Win, Cin), (N, H, W, Cout), dtype, True)


# ----------------------------------------------------------------------------- weights
def pack_weights(w_hwio: torch.Tensor, Cp: int, Kp: int, dtype, want_fwd=True, want_dgrad=True):
    R, S, Cr, Kr = w_hwio.shape
    assert w_hwio.dtype == torch.float32
    wf = torch.empty((Kp, R * S * Cp), dtype=dtype, device=w_hwio.device) if want_fwd else None
    wd = torch.empty((Cp, R * S * Kp), dtype=dtype, device=w_hwio.device) if want_dgrad else None
    A.check(A.lib().sgg_pack_conv_weights(_p(w_hwio), R, S, Cr, Kr, Cp, Kp, dt(dtype), _p(wf), _p(wd), _s()), "pack_conv_weights")
    return wf, wd


def pack_weights_batch(entries, dtype, device):
    """entries: [(w_hwio f32 view, Cp, Kp, want_fwd, want_dgrad)].  Allocates the packed buffers once and returns
    (table, n, max_elems, [(wf, wd)]): the device table of sgg_pack_item for repack_batch()."""
    items = (A.PackItem * len(entries))()
    bufs, max_elems = [], 0
    for it, (w, Cp, Kp, want_fwd, want_dgrad) in zip(items, entries):
        R, S, Cr, Kr = w.shape
        assert w.is_contiguous() and w.dtype == torch.float32
        wf = torch.empty((Kp, R * S * Cp), dtype=dtype, device=device) if want_fwd else None
        wd = torch.empty((Cp, R * S * Kp), dtype=dtype, device=device) if want_dgrad else None
        it.w, it.w_fwd, it.w_dgrad = w.data_ptr(), (wf.data_ptr() if wf is not None else None), (wd.data_ptr() if wd is not None else None)
        it.taps, it.C, it.K, it.Cpad, it.Kpad, it.reserved = R * S, Cr, Kr, Cp, Kp, 0
        bufs.append((wf, wd))
        max_elems = max(max_elems, R * S * Cp * Kp)
    raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(device)
    return raw, len(entries), max_elems, bufs


def repack_batch(table, n, max_elems, dtype):
    A.check(A.lib().sgg_pack_conv_weights_batch(_p(table), n, max_elems, dt(dtype), _s()), "pack_conv_weights_batch")


# ----------------------------------------------------------------------------- conv / deconv
def _out(out, shape, dtype, device):
    """The caller's output buffer (a contiguous slice of a stacked pair tensor) or a fresh one."""
    if out is None:
        return torch.empty(shape, dtype=dtype, device=device)
    assert tuple(out.shape) == tuple(shape) and out.dtype == dtype and out.is_contiguous()
    return out


def conv_fwd(g: ConvGeom, x, w_fwd, bias, act=A.ACT_NONE, leak=0.0, out=None):
    assert tuple(x.shape) == g.x_shape and not g.is_deconv, (tuple(x.shape), g.x_shape)
    y = _out(out, g.y_shape, x.dtype, x.device)
    pr = _prof("conv2d_fwd", g)
    if pr: pr.start()
    ws = workspace(g.ws_fwd, x.device) if g.ws_fwd else None
    A.check(A.lib().sgg_conv2d_fwd(C.byref(g.desc), _p(x), _p(w_fwd), _p(bias), _p(y), act, leak, _p(ws), g.ws_fwd, _s()), "conv2d_fwd")
    if pr: pr.stop()
    return y


def conv_fwd_stats(g: ConvGeom, x, w_fwd, bias, out=None, out_partial=None):
    """conv forward (no activation) + the per-chunk (sum, sumsq) rows of its output for the instance norm that follows."""
    assert tuple(x.shape) == g.x_shape and not g.is_deconv and g.stats_chunks > 0
    y = _out(out, g.y_shape, x.dtype, x.device)
    partial = _out(out_partial, (g.y_shape[0], g.stats_chunks, g.y_shape[3], 2), torch.float32, x.device)
    pr = _prof("conv2d_fwd", g)
    if pr: pr.start()
    ws = workspace(g.ws_fwd, x.device) if g.ws_fwd else None
    A.check(A.lib().sgg_conv2d_fwd_stats(C.byref(g.desc), _p(x), _p(w_fwd), _p(bias), _p(y), _p(partial), _p(ws), g.ws_fwd, _s()), "conv2d_fwd_stats")
    if pr: pr.stop()
    return y, partial


def conv_fwd_stats_pair(g: ConvGeom, x, w_fwd, bias, w_fwd2, bias2, nsplit):
    """conv_fwd_stats over a stacked batch of two networks: images [:nsplit] with (w_fwd, bias), the rest with (w_fwd2, bias2)."""
    assert tuple(x.shape) == g.x_shape and g.pair_ok and g.stats_chunks > 0 and 0 < nsplit < g.x_shape[0]
    y = torch.empty(g.y_shape, dtype=x.dtype, device=x.device)
    partial = torch.empty((g.y_shape[0], g.stats_chunks, g.y_shape[3], 2), dtype=torch.float32, device=x.device)
    pr = _prof("conv2d_fwd_pair", g)
    if pr: pr.start()
    ws = workspace(g.ws_fwd, x.device) if g.ws_fwd else None
    A.check(A.lib().sgg_conv2d_fwd_stats_pair(C.byref(g.desc), _p(x), _p(w_fwd), _p(bias), _p(w_fwd2), _p(bias2), nsplit, _p(y), _p(partial),
                                              _p(ws), g.ws_fwd, _s()), "conv2d_fwd_stats_pair")
    if pr: pr.stop()
    return y, partial


def conv_fwd_stats_normload(g: ConvGeom, x_raw, x_stats, gamma, beta, w_fwd, bias, pair=None):
    """conv(relu(instnorm(x_raw))) with the norm applied to the operand tiles inside the conv kernel ("normalise on load",
    sgg_conv2d_fwd_stats_normload).  x_stats: (mean, rstd) of x_raw (instnorm_finalize).  pair = (gamma2, beta2, w_fwd2,
    bias2, nsplit) for the lockstep pair.  Returns (x_norm, y, partial): the normalised operand (a by-product the backward
    pass needs), the conv output and its statistics rows -- all bit-identical to instnorm_fwd_partial + conv_fwd_stats."""
    assert tuple(x_raw.shape) == g.x_shape and g.normload_ok and g.stats_chunks > 0
    x_norm = torch.empty_like(x_raw)
    y = torch.empty(g.y_shape, dtype=x_raw.dtype, device=x_raw.device)
    partial = torch.empty((g.y_shape[0], g.stats_chunks, g.y_shape[3], 2), dtype=torch.float32, device=x_raw.device)
    gamma2, beta2, w2, bias2, nsplit = pair if pair is not None else (None, None, None, None, 0)
    pr = _prof("conv2d_fwd_normload" + ("_pair" if pair is not None else ""), g)
    if pr: pr.start()
    ws = workspace(g.ws_fwd, x_raw.device) if g.ws_fwd else None
    A.check(A.lib().sgg_conv2d_fwd_stats_normload(C.byref(g.desc), _p(x_raw), _p(x_stats), _p(gamma), _p(beta), _p(gamma2), _p(beta2),
                                                  _p(x_norm), _p(w_fwd), _p(bias), _p(w2), _p(bias2), nsplit, _p(y), _p(partial),
                                                  _p(ws), g.ws_fwd, _s()), "conv2d_fwd_stats_normload")
    if pr: pr.stop()
    return x_norm, y, partial


def instnorm_finalize(partial, HW, eps=1e-3):
    """(mean, rstd)[N][C] from a conv's statistics rows partial[N][chunks][C][2] (no pass over the tensor)."""
    N, chunks, Cp, _ = partial.shape
    stats = torch.empty((N, Cp, 2), dtype=torch.float32, device=partial.device)
    A.check(A.lib().sgg_instnorm_finalize(_p(partial), chunks, _p(stats), N, HW, Cp, eps, _s()), "instnorm_finalize")
    return stats


def conv_dgrad_pair(g: ConvGeom, dy, w_dgrad, w_dgrad2, nsplit, addend=None):
    assert tuple(dy.shape) == g.y_shape and g.pair_ok and 0 < nsplit < g.y_shape[0]
    assert addend is None or (tuple(addend.shape) == g.x_shape and addend.dtype == dy.dtype)
    dx = torch.empty(g.x_shape, dtype=dy.dtype, device=dy.device)
    pr = _prof("conv2d_bwd_data_pair", g)
    if pr: pr.start()
    ws = workspace(g.ws_dgrad, dy.device) if g.ws_dgrad else None
    A.check(A.lib().sgg_conv2d_bwd_data_pair(C.byref(g.desc), _p(dy), _p(w_dgrad), _p(w_dgrad2), nsplit, _p(addend), _p(dx), _p(ws), g.ws_dgrad, _s()),
            "conv2d_bwd_data_pair")
    if pr: pr.stop()
    return dx


def conv_dgrad(g: ConvGeom, dy, w_dgrad, addend=None, out_f32=False, out=None):
    """dx = conv^T(dy) (+ addend: the skip-connection gradient, fused into the epilogue).  out_f32 (mixed mode, needs
    g.dgrad_mixed): bf16 operands, dx in float32; the addend may then be bf16 or float32."""
    assert tuple(dy.shape) == g.y_shape and not g.is_deconv
    pr = _prof("conv2d_bwd_data", g)
    ws = workspace(g.ws_dgrad, dy.device) if g.ws_dgrad else None
    if out_f32:
        assert g.dgrad_mixed and dy.dtype == torch.bfloat16
        assert addend is None or (tuple(addend.shape) == g.x_shape and addend.dtype in (torch.bfloat16, torch.float32))
        dx = torch.empty(g.x_shape, dtype=torch.float32, device=dy.device)
        if pr: pr.start()
        A.check(A.lib().sgg_conv2d_bwd_data_mixed(C.byref(g.desc), _p(dy), _p(w_dgrad), _p(addend),
                                                  int(addend is not None and addend.dtype == torch.float32), _p(dx), _p(ws), g.ws_dgrad, _s()),
                "conv2d_bwd_data_mixed")
        if pr: pr.stop()
        return dx
    assert addend is None or (tuple(addend.shape) == g.x_shape and addend.dtype == dy.dtype)
    dx = _out(out, g.x_shape, dy.dtype, dy.device)
    if pr: pr.start()
    A.check(A.lib().sgg_conv2d_bwd_data(C.byref(g.desc), _p(dy), _p(w_dgrad), _p(addend), _p(dx), _p(ws), g.ws_dgrad, _s()), "conv2d_bwd_data")
    if pr: pr.stop()
    return dx


def conv_dgrad_stats(g: ConvGeom, dy, w_dgrad, addend, norm_x, norm_stats, norm_gamma, norm_beta, norm_act=A.ACT_NONE, norm_leak=0.0):
    """conv_dgrad + the first pass of the instance-norm backward that consumes dx (the norm whose input is norm_x)."""
    assert tuple(dy.shape) == g.y_shape and not g.is_deconv and g.bwd_stats_chunks > 0
    assert tuple(norm_x.shape) == g.x_shape and norm_x.dtype == dy.dtype
    assert addend is None or (tuple(addend.shape) == g.x_shape and addend.dtype == dy.dtype)
    dx = torch.empty(g.x_shape, dtype=dy.dtype, device=dy.device)
    partial = torch.empty((g.x_shape[0], g.bwd_stats_chunks, g.x_shape[3], 2), dtype=torch.float32, device=dy.device)
    pr = _prof("conv2d_bwd_data", g)
    if pr: pr.start()
    ws = workspace(g.ws_dgrad, dy.device) if g.ws_dgrad else None
    A.check(A.lib().sgg_conv2d_bwd_data_stats(C.byref(g.desc), _p(dy), _p(w_dgrad), _p(addend), _p(dx), _p(norm_x), _p(norm_stats),
                                              _p(norm_gamma), _p(norm_beta), norm_act, norm_leak, _p(partial), _p(ws), g.ws_dgrad, _s()),
            "conv2d_bwd_data_stats")
    if pr: pr.stop()
    return dx, partial


def conv_wgrad(g: ConvGeom, x, dy, dw, accumulate=False):
    """dw: f32 (R,S,C_real,K_real) view to write / accumulate into."""
    assert tuple(x.shape) == g.x_shape and tuple(dy.shape) == g.y_shape and dw.dtype == torch.float32
    ws = workspace(g.ws_wgrad, x.device)
    pr = _prof("conv2d_bwd_weight", g)
    if pr: pr.start()
    A.check(A.lib().sgg_conv2d_bwd_weight(C.byref(g.desc), _p(x), _p(dy), _p(dw), dw.shape[2], dw.shape[3], int(accumulate),
                                          _p(ws), ws.numel(), _s()), "conv2d_bwd_weight")
    if pr: pr.stop()


def conv_wgrad_pair(g: ConvGeom, x0, dy0, x1, dy1, dw, accumulate=False):
    """dw (+)= wgrad(x0, dy0) + wgrad(x1, dy1): two applications of one layer, one launch (needs g.wgrad_pair)."""
    assert g.wgrad_pair and dw.dtype == torch.float32
    for x, dy in ((x0, dy0), (x1, dy1)):
        assert tuple(x.shape) == g.x_shape and tuple(dy.shape) == g.y_shape
    ws = workspace(g.ws_wgrad, x0.device)
    pr = _prof("conv2d_bwd_weight_pair", g)
    if pr: pr.start()
    A.check(A.lib().sgg_conv2d_bwd_weight_pair(C.byref(g.desc), _p(x0), _p(dy0), _p(x1), _p(dy1), _p(dw), dw.shape[2], dw.shape[3],
                                               int(accumulate), _p(ws), ws.numel(), _s()), "conv2d_bwd_weight_pair")
    if pr: pr.stop()


def conv_wgrad_pair2(g: ConvGeom, net_a, net_b, accumulate=False):
    """Two networks of one architecture, each applied twice: net = (x0, dy0, x1, dy1, dw); ONE launch, half the slabs per network.
    Returns False (nothing launched) when the shape has no such form."""
    assert g.wgrad_pair
    dwa, dwb = net_a[4], net_b[4]
    assert dwa.dtype == torch.float32 and dwb.dtype == torch.float32 and dwa.shape == dwb.shape
    for x0, d0, x1, d1, _ in (net_a, net_b):
        for x, dy in ((x0, d0), (x1, d1)):
            assert tuple(x.shape) == g.x_shape and tuple(dy.shape) == g.y_shape
    ws = workspace(g.ws_wgrad, dwa.device)
    pr = _prof("conv2d_bwd_weight_pair2", g)
    if pr: pr.start()
    rc = A.lib().sgg_conv2d_bwd_weight_pair2(C.byref(g.desc), *[_p(t) for t in net_a], *[_p(t) for t in net_b], dwa.shape[2], dwa.shape[3],
                                             int(accumulate), _p(ws), ws.numel(), _s())
    if rc == A.EUNSUPPORTED:
        if pr: pr.stop()
        return False
    A.check(rc, "conv2d_bwd_weight_pair2")
    if pr: pr.stop()
    return True


# ---- grouped launches (sgg_*_group2): the same call site of two networks in one call.  `g` is ONE network's geometry; the tensors
# hold 2N images, the first network's first.  Bit-identical to two single calls.
def _stacked(shape):
    return (2 * shape[0],) + tuple(shape[1:])


def conv_fwd_group2(g: ConvGeom, x, w_fwd, bias, w_fwd2, bias2, act=A.ACT_NONE, leak=0.0, out=None):
    assert tuple(x.shape) == _stacked(g.x_shape) and not g.is_deconv
    y = _out(out, _stacked(g.y_shape), x.dtype, x.device)
    ws = workspace(2 * g.ws_fwd, x.device) if g.ws_fwd else None
    pr = _prof("conv2d_fwd_group2", g)
    if pr: pr.start()
    A.check(A.lib().sgg_conv2d_fwd_group2(C.byref(g.desc), _p(x), _p(w_fwd), _p(bias), _p(w_fwd2), _p(bias2), _p(y), act, leak,
                                          _p(ws), 2 * g.ws_fwd, _s()), "conv2d_fwd_group2")
    if pr: pr.stop()
    return y


def conv_dgrad_group2(g: ConvGeom, dy, w_dgrad, w_dgrad2, addend=None):
    assert tuple(dy.shape) == _stacked(g.y_shape) and not g.is_deconv
    assert addend is None or (tuple(addend.shape) == _stacked(g.x_shape) and addend.dtype == dy.dtype)
    dx = torch.empty(_stacked(g.x_shape), dtype=dy.dtype, device=dy.device)
    ws = workspace(2 * g.ws_dgrad, dy.device) if g.ws_dgrad else None
    pr = _prof("conv2d_bwd_data_group2", g)
    if pr: pr.start()
    A.check(A.lib().sgg_conv2d_bwd_data_group2(C.byref(g.desc), _p(dy), _p(w_dgrad), _p(w_dgrad2), _p(addend), _p(dx),
                                               _p(ws), 2 * g.ws_dgrad, _s()), "conv2d_bwd_data_group2")
    if pr: pr.stop()
    return dx


def conv_wgrad_group2(g: ConvGeom, x, dy, dw, dw2, accumulate=False):
    """dw (+)= wgrad(x[:N], dy[:N]), dw2 (+)= wgrad(x[N:], dy[N:]): both networks' slabs, one reduce launch."""
    assert tuple(x.shape) == _stacked(g.x_shape) and tuple(dy.shape) == _stacked(g.y_shape) and dw.shape == dw2.shape
    ws = workspace(2 * g.ws_wgrad, x.device)
    fn = A.lib().sgg_deconv2d_bwd_weight_group2 if g.is_deconv else A.lib().sgg_conv2d_bwd_weight_group2
    pr = _prof("bwd_weight_group2", g)
    if pr: pr.start()
    A.check(fn(C.byref(g.desc), _p(x), _p(dy), _p(dw), _p(dw2), dw.shape[2], dw.shape[3], int(accumulate), _p(ws), ws.numel(), _s()),
            "bwd_weight_group2")
    if pr: pr.stop()


def deconv_fwd_group2(g: ConvGeom, x, w_dgrad, bias, w_dgrad2, bias2, act=A.ACT_NONE, leak=0.0):
    assert tuple(x.shape) == _stacked(g.x_shape) and g.is_deconv
    y = torch.empty(_stacked(g.y_shape), dtype=x.dtype, device=x.device)
    ws = workspace(2 * g.ws_fwd, x.device) if g.ws_fwd else None
    pr = _prof("deconv2d_fwd_group2", g)
    if pr: pr.start()
    A.check(A.lib().sgg_deconv2d_fwd_group2(C.byref(g.desc), _p(x), _p(w_dgrad), _p(bias), _p(w_dgrad2), _p(bias2), _p(y), act, leak,
                                            _p(ws), 2 * g.ws_fwd, _s()), "deconv2d_fwd_group2")
    if pr: pr.stop()
    return y


def deconv_dgrad_group2(g: ConvGeom, dy, w_fwd, w_fwd2):
    assert tuple(dy.shape) == _stacked(g.y_shape) and g.is_deconv
    dx = torch.empty(_stacked(g.x_shape), dtype=dy.dtype, device=dy.device)
    ws = workspace(2 * g.ws_dgrad, dy.device) if g.ws_dgrad else None
    pr = _prof("deconv2d_bwd_data_group2", g)
    if pr: pr.start()
    A.check(A.lib().sgg_deconv2d_bwd_data_group2(C.byref(g.desc), _p(dy), _p(w_fwd), _p(w_fwd2), _p(dx), _p(ws), 2 * g.ws_dgrad, _s()),
            "deconv2d_bwd_data_group2")
    if pr: pr.stop()
    return dx


def deconv_fwd(g: ConvGeom, x, w_dgrad, bias, act=A.ACT_NONE, leak=0.0, out=None):
    assert tuple(x.shape) == g.x_shape and g.is_deconv
    y = _out(out, g.y_shape, x.dtype, x.device)
    ws = workspace(g.ws_fwd, x.device) if g.ws_fwd else None
    A.check(A.lib().sgg_deconv2d_fwd(C.byref(g.desc), _p(x), _p(w_dgrad), _p(bias), _p(y), act, leak, _p(ws), g.ws_fwd, _s()), "deconv2d_fwd")
    return y


def deconv_fwd_stats(g: ConvGeom, x, w_dgrad, bias, pair=None):
    """Conv2DTranspose forward (no activation) + the per-chunk (sum, sumsq) rows of its output for the instance norm behind it.
    pair = (w_dgrad2, bias2, nsplit): `g` describes a stacked batch of two networks, images >= nsplit use the second weight set."""
    assert tuple(x.shape) == g.x_shape and g.is_deconv and g.stats_chunks > 0
    y = torch.empty(g.y_shape, dtype=x.dtype, device=x.device)
    partial = torch.empty((g.y_shape[0], g.stats_chunks, g.y_shape[3], 2), dtype=torch.float32, device=x.device)
    w2, b2, ns = pair if pair is not None else (None, None, 0)
    pr = _prof("deconv2d_fwd_stats", g)
    if pr: pr.start()
    A.check(A.lib().sgg_deconv2d_fwd_stats(C.byref(g.desc), _p(x), _p(w_dgrad), _p(bias), _p(w2), _p(b2), int(ns), _p(y), _p(partial),
                                           None, 0, _s()), "deconv2d_fwd_stats")
    if pr: pr.stop()
    return y, partial


def deconv_dgrad(g: ConvGeom, dy, w_fwd, out=None):
    assert tuple(dy.shape) == g.y_shape and g.is_deconv
    dx = _out(out, g.x_shape, dy.dtype, dy.device)
    ws = workspace(g.ws_dgrad, dy.device) if g.ws_dgrad else None
    A.check(A.lib().sgg_deconv2d_bwd_data(C.byref(g.desc), _p(dy), _p(w_fwd), _p(dx), _p(ws), g.ws_dgrad, _s()), "deconv2d_bwd_data")
    return dx


def deconv_wgrad(g: ConvGeom, x, dy, dw, accumulate=False):
    """dw: f32 (R,S,Cout_real,Cin_real) -- the Keras transpose-kernel layout."""
    assert tuple(x.shape) == g.x_shape and tuple(dy.shape) == g.y_shape
    ws = workspace(g.ws_wgrad, x.device)
    A.check(A.lib().sgg_deconv2d_bwd_weight(C.byref(g.desc), _p(x), _p(dy), _p(dw), dw.shape[2], dw.shape[3], int(accumulate),
                                            _p(ws), ws.numel(), _s()), "deconv2d_bwd_weight")


def bias_grad(dy, db, accumulate=False):
    """db[c] (+)= sum over all pixels of dy[..., c] for c < db.numel()."""
    Cp = dy.shape[-1]
    P = dy.numel() // Cp
    need = int(A.lib().sgg_bias_grad_workspace(P, Cp))
    ws = workspace(need, dy.device)
    A.check(A.lib().sgg_bias_grad(_p(dy), _p(db), P, Cp, db.numel(), int(accumulate), dt(dy), _p(ws), ws.numel(), _s()), "bias_grad")


def bias_grad_group2(dy, db, db2, accumulate=False):
    """Two networks' tensors stacked on the batch dimension: db (+)= colsum(dy[:N]), db2 (+)= colsum(dy[N:])."""
    Cp = dy.shape[-1]
    P = dy.numel() // Cp // 2
    need = 2 * int(A.lib().sgg_bias_grad_workspace(P, Cp))
    ws = workspace(need, dy.device)
    pr = _prof("bias_grad_group2", (P, Cp))
    if pr: pr.start()
    A.check(A.lib().sgg_bias_grad_group2(_p(dy), _p(db), _p(db2), P, Cp, db.numel(), int(accumulate), dt(dy), _p(ws), ws.numel(), _s()), "bias_grad_group2")
    if pr: pr.stop()


# ----------------------------------------------------------------------------- instance norm / activations
def instnorm_fwd(x, gamma, beta, residual=None, eps=1e-3, act=A.ACT_NONE, leak=0.0):
    N, H, W, Cp = x.shape
    assert gamma.numel() == Cp and beta.numel() == Cp, "gamma/beta must be channel-padded"
    y = torch.empty_like(x)
    stats = torch.empty((N, Cp, 2), dtype=torch.float32, device=x.device)
    ws = workspace(int(A.lib().sgg_instnorm_workspace(N, H * W, Cp)), x.device)
    pr = _prof("instnorm_fwd", (tuple(x.shape), residual is not None))
    if pr: pr.start()
    A.check(A.lib().sgg_instnorm_fwd(_p(x), _p(gamma), _p(beta), _p(residual), _p(y), _p(stats), N, H * W, Cp, eps, act, leak,
                                     dt(x), _p(ws), ws.numel(), _s()), "instnorm_fwd")
    if pr: pr.stop()
    return y, stats


def instnorm_fwd_partial(x, partial, gamma, beta, residual=None, eps=1e-3, act=A.ACT_NONE, leak=0.0):
    """instance norm whose statistics pass was done by the producing conv (conv_fwd_stats): finalize + apply only."""
    N, H, W, Cp = x.shape
    assert gamma.numel() == Cp and beta.numel() == Cp and partial.shape[0] == N and partial.shape[2] == Cp
    y = torch.empty_like(x)
    stats = torch.empty((N, Cp, 2), dtype=torch.float32, device=x.device)
    pr = _prof("instnorm_fwd_partial", (tuple(x.shape), residual is not None))
    if pr: pr.start()
    A.check(A.lib().sgg_instnorm_fwd_partial(_p(x), _p(gamma), _p(beta), _p(residual), _p(y), _p(stats), _p(partial), partial.shape[1],
                                             N, H * W, Cp, eps, act, leak, dt(x), _s()), "instnorm_fwd_partial")
    if pr: pr.stop()
    return y, stats


def instnorm_bwd(dy, x, gamma, beta, stats, dgamma, dbeta, accumulate=False, act=A.ACT_NONE, leak=0.0):
    N, H, W, Cp = x.shape
    dx = torch.empty_like(x)
    ws = workspace(int(A.lib().sgg_instnorm_workspace(N, H * W, Cp)), x.device)
    if dy.dtype == torch.float32 and x.dtype == torch.bfloat16:        # mixed mode: f32 gradient chain, bf16 tensors
        A.check(A.lib().sgg_instnorm_bwd_mixed(_p(dy), _p(x), _p(gamma), _p(beta), _p(stats), _p(dx), _p(dgamma), _p(dbeta), N, H * W, Cp,
                                               dgamma.numel(), int(accumulate), act, leak, _p(ws), ws.numel(), _s()), "instnorm_bwd_mixed")
        return dx
    A.check(A.lib().sgg_instnorm_bwd(_p(dy), _p(x), _p(gamma), _p(beta), _p(stats), _p(dx), _p(dgamma), _p(dbeta), N, H * W, Cp,
                                     dgamma.numel(), int(accumulate), act, leak, dt(x), _p(ws), ws.numel(), _s()), "instnorm_bwd")
    return dx


def instnorm_bwd_partial(dy, x, partial, gamma, beta, stats, dgamma, dbeta, accumulate=False, act=A.ACT_NONE, leak=0.0):
    """instance-norm backward whose statistics pass was done by the conv data gradient that produced dy."""
    N, H, W, Cp = x.shape
    assert partial.shape[0] == N and partial.shape[2] == Cp
    dx = torch.empty_like(x)
    ws = workspace(N * Cp * 16, x.device)
    A.check(A.lib().sgg_instnorm_bwd_partial(_p(dy), _p(x), _p(gamma), _p(beta), _p(stats), _p(dx), _p(dgamma), _p(dbeta), _p(partial),
                                             partial.shape[1], N, H * W, Cp, dgamma.numel(), int(accumulate), act, leak, dt(x),
                                             _p(ws), ws.numel(), _s()), "instnorm_bwd_partial")
    return dx


# ---- two networks of the same shape on one stacked batch (images 0..nsplit-1: first network; the rest: second)
def instnorm_fwd_pair(x, gamma, beta, gamma2, beta2, nsplit, residual=None, eps=1e-3, act=A.ACT_NONE, leak=0.0):
    N, H, W, Cp = x.shape
    assert gamma.numel() == Cp and gamma2.numel() == Cp and 0 < nsplit < N
    y = torch.empty_like(x)
    stats = torch.empty((N, Cp, 2), dtype=torch.float32, device=x.device)
    ws = workspace(int(A.lib().sgg_instnorm_workspace(N, H * W, Cp)), x.device)
    pr = _prof("instnorm_fwd_pair", (tuple(x.shape), residual is not None))
    if pr: pr.start()
    A.check(A.lib().sgg_instnorm_fwd_pair(_p(x), _p(gamma), _p(beta), _p(gamma2), _p(beta2), nsplit, _p(residual), _p(y), _p(stats), N, H * W, Cp,
                                          eps, act, leak, dt(x), _p(ws), ws.numel(), _s()), "instnorm_fwd_pair")
    if pr: pr.stop()
    return y, stats


def instnorm_fwd_partial_pair(x, partial, gamma, beta, gamma2, beta2, nsplit, residual=None, eps=1e-3, act=A.ACT_NONE, leak=0.0):
    N, H, W, Cp = x.shape
    assert gamma.numel() == Cp and gamma2.numel() == Cp and partial.shape[0] == N and partial.shape[2] == Cp and 0 < nsplit < N
    y = torch.empty_like(x)
    stats = torch.empty((N, Cp, 2), dtype=torch.float32, device=x.device)
    pr = _prof("instnorm_fwd_partial_pair", (tuple(x.shape), residual is not None))
    if pr: pr.start()
    A.check(A.lib().sgg_instnorm_fwd_partial_pair(_p(x), _p(gamma), _p(beta), _p(gamma2), _p(beta2), nsplit, _p(residual), _p(y), _p(stats),
                                                  _p(partial), partial.shape[1], N, H * W, Cp, eps, act, leak, dt(x), _s()), "instnorm_fwd_partial_pair")
    if pr: pr.stop()
    return y, stats


def instnorm_bwd_pair(dy, x, gamma, beta, gamma2, beta2, nsplit, stats, dgamma, dbeta, dgamma2, dbeta2, accumulate=False,
                      act=A.ACT_NONE, leak=0.0):
    N, H, W, Cp = x.shape
    assert dy.dtype == x.dtype and dgamma.numel() == dgamma2.numel() and 0 < nsplit < N
    dx = torch.empty_like(x)
    ws = workspace(int(A.lib().sgg_instnorm_workspace(N, H * W, Cp)), x.device)
    A.check(A.lib().sgg_instnorm_bwd_pair(_p(dy), _p(x), _p(gamma), _p(beta), _p(gamma2), _p(beta2), nsplit, _p(stats), _p(dx), _p(dgamma), _p(dbeta),
                                          _p(dgamma2), _p(dbeta2), N, H * W, Cp, dgamma.numel(), int(accumulate), act, leak, dt(x),
                                          _p(ws), ws.numel(), _s()), "instnorm_bwd_pair")
    return dx


def act_fwd(x, act, leak=0.0):
    y = torch.empty_like(x)
    A.check(A.lib().sgg_act_fwd(_p(x), _p(y), x.numel(), act, leak, dt(x), _s()), "act_fwd")
    return y


def act_bwd(dy, y, act, leak=0.0):
    dx = torch.empty_like(dy)
    A.check(A.lib().sgg_act_bwd(_p(dy), _p(y), _p(dx), dy.numel(), act, leak, dt(dy), _s()), "act_bwd")
    return dx


def add(a, b):
    o = torch.empty_like(a)
    A.check(A.lib().sgg_add(_p(a), _p(b), _p(o), a.numel(), dt(a), _s()), "add")
    return o


# ----------------------------------------------------------------------------- mask / losses / optimizer
def mask_reduce_fwd(h4, mask, C_real):
    N, hh, hw, Cp = h4.shape
    _, mh, mw, Cm = mask.shape
    assert Cm == C_real and mask.dtype == torch.float32
    out = torch.empty((N, mh, mw, 1), dtype=torch.float32, device=h4.device)
    A.check(A.lib().sgg_mask_reduce_fwd(_p(h4), _p(mask), _p(out), N, hh, hw, mh, mw, C_real, Cp, dt(h4), _s()), "mask_reduce_fwd")
    return out


def mask_reduce_bwd(dout, mask, h4_shape, dtype, C_real):
    N, hh, hw, Cp = h4_shape
    _, mh, mw, _ = mask.shape
    dh4 = torch.empty(h4_shape, dtype=dtype, device=dout.device)
    A.check(A.lib().sgg_mask_reduce_bwd(_p(dout), _p(mask), _p(dh4), N, hh, hw, mh, mw, C_real, Cp, dt(dtype), _s()), "mask_reduce_bwd")
    return dh4


def bce_logits(logits, label, loss, dlogits=None, weight=1.0, gscale=1.0, accumulate_loss=False, accumulate_grad=False):
    """loss: 1-element f32 tensor written/accumulated on device (no host sync)."""
    assert logits.dtype == torch.float32
    A.check(A.lib().sgg_bce_logits(_p(logits), logits.numel(), float(label), float(weight), float(gscale), _p(loss), _p(dlogits),
                                   int(bool(accumulate_loss)) | (int(bool(accumulate_grad)) << 1), _s()), "bce_logits")


def l1_loss(a, b, C_real, loss, db=None, weight=1.0, gscale=1.0, accumulate=False, accumulate_grad=False):
    Cp = a.shape[-1]
    P = a.numel() // Cp
    ws = workspace(int(A.lib().sgg_l1_loss_workspace(P, Cp)), a.device)
    A.check(A.lib().sgg_l1_loss(_p(a), _p(b), P, C_real, Cp, float(weight), float(gscale), _p(loss), _p(db),
                                int(bool(accumulate)) | (int(bool(accumulate_grad)) << 1), dt(a), _p(ws), ws.numel(), _s()), "l1_loss")


def mse_const(x, target, loss, dx=None, weight=1.0, gscale=1.0, accumulate_loss=False, accumulate_grad=False):
    """mae_criterion(x, target*ones) (module.py:340-341): mean squared error against a constant."""
    assert x.dtype == torch.float32
    A.check(A.lib().sgg_mse_const(_p(x), x.numel(), float(target), float(weight), float(gscale), _p(loss), _p(dx),
                                  int(bool(accumulate_loss)) | (int(bool(accumulate_grad)) << 1), _s()), "mse_const")


def seg_edge_weight(seg, C_real):
    """model.py:108-119 weighted_seg: (N,H,W,Cp) colour segmentation -> f32 (N,H,W) edge indicator."""
    N, H, W, Cp = seg.shape
    out = torch.empty((N, H, W), dtype=torch.float32, device=seg.device)
    A.check(A.lib().sgg_seg_edge_weight(_p(seg), _p(out), N, H, W, C_real, Cp, dt(seg), _s()), "seg_edge_weight")
    return out


def gradloss(x, target, weight, C_real, loss, dx=None, lam=1.0, gscale=1.0, accumulate_loss=False, accumulate_grad=False):
    """gradloss_criterion (module.py:347-351) scaled by lam; optional gradient w.r.t. x."""
    N, H, W, Cp = x.shape
    need = int(A.lib().sgg_gradloss_workspace(N, H, W, C_real))
    ws = workspace(need, x.device)
    A.check(A.lib().sgg_gradloss(_p(x), _p(target), _p(weight), N, H, W, C_real, Cp, float(lam), float(gscale), _p(loss), _p(dx),
                                 int(bool(accumulate_loss)) | (int(bool(accumulate_grad)) << 1), dt(x), _p(ws), ws.numel(), _s()), "gradloss")


def adam(theta, g, m, v, t, lr=1e-3, beta1=0.5, beta2=0.999, eps=1e-7, grad_scale=1.0):
    assert theta.dtype == torch.float32 and theta.numel() == g.numel() == m.numel() == v.numel()
    A.check(A.lib().sgg_adam(_p(theta), _p(g), _p(m), _p(v), theta.numel(), int(t), lr, beta1, beta2, eps, grad_scale, _s()), "adam")


def adam_iter(theta, g, m, v, state, lr=1e-3, beta1=0.5, beta2=0.999, eps=1e-7, grad_scale=1.0):
    """Adam with the step number kept on the device (``state``: int64[2] = [iterations, scratch]; iterations is incremented)."""
    assert theta.dtype == torch.float32 and theta.numel() == g.numel() == m.numel() == v.numel()
    assert state.dtype == torch.int64 and state.numel() == 2
    A.check(A.lib().sgg_adam_iter(_p(theta), _p(g), _p(m), _p(v), theta.numel(), _p(state), lr, beta1, beta2, eps, grad_scale, _s()), "adam_iter")


# ----------------------------------------------------------------------------- data side
def seg_class_map(rgb_u8):
    """uint8 (..., M, N, 3|4) -> uint8 (..., M, N) class indices (segment_class.py:60-99), bit exact."""
    assert rgb_u8.dtype == torch.uint8 and rgb_u8.shape[-1] >= 3
    out = torch.empty(rgb_u8.shape[:-1], dtype=torch.uint8, device=rgb_u8.device)
    A.check(A.lib().sgg_seg_class_map(_p(rgb_u8), rgb_u8.shape[-1], out.numel(), _p(out), _s()), "seg_class_map")
    return out


def onehot_resample(idx_u8, oh, ow, n_classes):
    N, H, W = idx_u8.shape
    mask = torch.empty((N, oh, ow, n_classes), dtype=torch.float32, device=idx_u8.device)
    A.check(A.lib().sgg_onehot_resample(_p(idx_u8), _p(mask), N, H, W, oh, ow, n_classes, _s()), "onehot_resample")
    return mask


def pad_channels(x_f32, Cd, dtype, out=None):
    Cs = x_f32.shape[-1]
    out = _out(out, tuple(x_f32.shape[:-1]) + (Cd,), dtype, x_f32.device)
    A.check(A.lib().sgg_pad_channels(_p(x_f32), _p(out), x_f32.numel() // Cs, Cs, Cd, dt(dtype), _s()), "pad_channels")
    return out


def unpad_channels(x, Cd):
    Cs = x.shape[-1]
    out = torch.empty(x.shape[:-1] + (Cd,), dtype=torch.float32, device=x.device)
    A.check(A.lib().sgg_unpad_channels(_p(x), _p(out), x.numel() // Cs, Cs, Cd, dt(x), _s()), "unpad_channels")
    return out
