"""GPU parity tests, op level: every HIP kernel behind the C ABI vs the CPU oracle on seeded inputs.

float32 path: rtol 1e-4 (fp32 kernels vs fp64 oracle); bfloat16 path: 2e-2 of the
tensor's scale; integer paths: bit exact.
"""
import os

import numpy as np
import pytest
import torch

from oracle import sggan_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def sg():
    import sggan_amd
    import sggan_amd.kernels  # noqa: F401
    import sggan_amd.segment_class  # noqa: F401
    assert os.path.exists(sggan_amd.LIB_PATH)
    return sggan_amd


def V(a):
    return O.Var(np.asarray(a, np.float64))


def close(got, exp, dtype, what="", scale=None):
    got = np.asarray(got, np.float64)
    exp = np.asarray(exp, np.float64)
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    s = scale if scale is not None else max(np.abs(exp).max(), 1e-6)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    err = np.abs(got - exp).max() / s
    assert err < tol, f"{what}: max err {err:.3e} of scale {s:.3e} (tol {tol})"


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a, np.float32)).to("cuda").to(dtype)


DT = [torch.float32, torch.bfloat16]

# (name, R, stride, padding, Cin, Cout, H, W)
CONV_CASES = [
    ("res3x3_reflect", 3, 1, "REFLECT-1", 16, 24, 9, 11),
    ("res3x3_reflect_big", 3, 1, "REFLECT-1", 128, 128, 12, 10),
    ("stem7x7_reflect", 7, 1, "REFLECT-3", 3, 16, 12, 13),
    ("head7x7_reflect", 7, 1, "REFLECT-3", 16, 3, 10, 9),
    ("same_s2_even", 3, 2, "SAME", 8, 16, 12, 16),
    ("same_s2_odd", 3, 2, "SAME", 16, 8, 11, 13),
    ("same_s1", 3, 1, "SAME", 32, 34, 5, 13),
    ("valid_s2_odd", 3, 2, "VALID", 16, 64, 15, 31),
    ("valid_s2_even", 3, 2, "VALID", 8, 8, 16, 32),
    ("valid_s1", 3, 1, "VALID", 64, 72, 7, 15),
    ("d_h0", 3, 2, "SAME", 3, 64, 16, 16),
    ("wide_k", 3, 1, "SAME", 256, 136, 6, 6),
    ("res_full_width", 3, 1, "REFLECT-1", 256, 256, 10, 12),
    ("d_h3_like", 3, 1, "SAME", 256, 512, 6, 9),
    ("c3_like_s2", 3, 2, "SAME", 128, 256, 12, 10),
    ("h31_like", 3, 2, "VALID", 512, 512, 9, 11),
    ("head_halo_7x7", 7, 1, "REFLECT-3", 64, 3, 32, 64),
    ("head_halo_3x3_same", 3, 1, "SAME", 128, 10, 16, 32),
    ("stem_halo_7x7", 7, 1, "REFLECT-3", 3, 64, 32, 64),
    ("stem_halo_3x3_same", 3, 1, "SAME", 8, 64, 16, 32),
    ("head_dgrad_halo_same", 5, 1, "SAME", 64, 8, 16, 64),
    ("halo3_reflect", 3, 1, "REFLECT-1", 64, 256, 4, 128),        # LDS-resident 3x3 halo GEMM (bf16), one chunk
    ("halo3_reflect_chunks", 3, 1, "REFLECT-1", 192, 320, 4, 128),  # 3 / 5 chunks (halo refill schedule), 2 channel tiles
    ("halo3_reflect_wide", 3, 1, "REFLECT-1", 64, 64, 6, 256),    # two column tiles: left / right mirror patches apart
    ("halo3_same_tail", 3, 1, "SAME", 128, 192, 6, 256),          # zero padding fwd + dgrad, column tiles, tails
    ("s2halo_c2_like", 3, 2, "SAME", 64, 128, 16, 64),            # stride-2 data gradient, all parity classes per block
    ("s2halo_c3_like", 3, 2, "SAME", 128, 192, 32, 128),          # two output-channel tiles, 3 dy chunks, 2x2 pixel tiles
    ("w9s2_c2_like", 3, 2, "SAME", 64, 128, 8, 128),              # stride-2 all-taps weight gradient (parity-split halo)
    ("w9s2_two_ctiles", 3, 2, "SAME", 128, 128, 4, 256),          # two input-channel tiles, two column tiles per row
    ("w9_reflect", 3, 1, "REFLECT-1", 64, 128, 4, 64),            # all-taps halo wgrad (bf16): one output tile, 4 pixel tiles
    ("w9_same_multi", 3, 1, "SAME", 128, 256, 6, 128),            # zero padding, 2x2 output tiles, column tiles
]


@pytest.mark.parametrize("dtype", DT, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_fwd_bwd(sg, case, dtype):
    _, R, stride, padding, Ci, Co, H, W = case
    import zlib
    rng = np.random.default_rng(zlib.crc32(case[0].encode()))
    N = 2
    q = (lambda a: dev(a, dtype).detach().float().cpu().numpy().astype(np.float64))      # inputs rounded to the storage dtype
    x = q(rng.standard_normal((N, H, W, Ci)))
    w = q(rng.standard_normal((R, R, Ci, Co)) / np.sqrt(R * R * Ci))
    b = rng.standard_normal(Co).astype(np.float32).astype(np.float64)
    pad, refl = ("VALID", int(padding.split("-")[1])) if padding.startswith("REFLECT") else (padding, 0)
    t = O.Tape()
    vx, vw, vb = V(x), V(w), V(b)
    y = O.conv2d(t, vx, vw, vb, stride, pad, refl)
    dy = q(rng.standard_normal(y.v.shape))
    t.backward([(y, dy)])

    tx = dev(x, dtype).requires_grad_(True)
    tw = dev(w).requires_grad_(True)
    tb = dev(b).requires_grad_(True)
    ty = sg.conv2d(tx, tw, tb, stride=stride, padding=padding)
    close(ty.detach().float().cpu().numpy(), y.v, dtype, "y")
    ty.backward(dev(dy, dtype))
    close(tx.grad.detach().float().cpu().numpy(), vx.g, dtype, "dx")
    close(tw.grad.detach().cpu().numpy(), vw.g, dtype, "dw")
    close(tb.grad.detach().cpu().numpy(), vb.g, dtype, "db")


@pytest.mark.parametrize("shape", [(2, 4, 128, 64, 256), (1, 6, 256, 128, 320)], ids=["one_tile_col", "two_tile_cols_ktail"])
def test_conv_fwd_stats_epilogue(sg, shape):
    """conv forward + instance-norm statistics epilogue (sgg_conv2d_fwd_stats): the output is bit-identical to the plain
    forward, the per-chunk (sum, sumsq) rows add up to the sums of the STORED bf16 output, and the norm fed with them
    matches the norm that makes its own statistics pass."""
    from sggan_amd import kernels as K, _abi as A
    N, H, W, Ci, Co = shape
    rng = np.random.default_rng(3)
    x = dev(rng.standard_normal((N, H, W, Ci)), torch.bfloat16)
    w = dev(rng.standard_normal((3, 3, Ci, Co)) / np.sqrt(9 * Ci))
    b = dev(rng.standard_normal(Co))
    g = K.conv_geom(N, H, W, Ci, Co, 3, 3, 1, "VALID", 1, torch.bfloat16)
    assert g.stats_chunks == (H // 2) * (W // 128) * 2
    wf, _ = K.pack_weights(w, Ci, Co, torch.bfloat16)
    y0 = K.conv_fwd(g, x, wf, b)
    y1, part = K.conv_fwd_stats(g, x, wf, b)
    assert torch.equal(y0, y1)
    yf = y1.float().cpu().numpy().astype(np.float64)
    got = part.cpu().numpy().astype(np.float64).sum(axis=1)                   # (N, Co, 2)
    s1, s2 = yf.sum(axis=(1, 2)), (yf * yf).sum(axis=(1, 2))
    assert np.abs(got[..., 0] - s1).max() < 1e-4 * max(1.0, np.abs(s1).max()) + 1e-2
    assert np.abs(got[..., 1] - s2).max() < 1e-4 * s2.max()
    gam, bet = dev(1 + 0.2 * rng.standard_normal(Co)), dev(0.2 * rng.standard_normal(Co))
    za, sa = K.instnorm_fwd(y1, gam, bet, None, 1e-3, A.ACT_RELU)
    zb, sb = K.instnorm_fwd_partial(y1, part, gam, bet, None, 1e-3, A.ACT_RELU)
    assert (sa - sb).abs().max().item() < 1e-4
    assert (za.float() - zb.float()).abs().max().item() < 2e-2
    # shapes without the epilogue report 0 chunks and the entry point refuses them
    g2 = K.conv_geom(N, 5, 7, Ci, Co, 3, 3, 1, "VALID", 1, torch.bfloat16)
    assert g2.stats_chunks == 0


def test_stem_conv_fwd_stats_epilogue(sg):
    """The generator stem (7x7 REFLECT, 3(8) -> 64 channels, module.py:230-233) through sgg_conv2d_fwd_stats: the narrow-input kernel's
    16-byte-store epilogue also emits one (sum, sumsq) row per (image, 16 x 32 tile).  Output bit-identical to the plain forward; rows add
    up to the sums of the stored output; the norm fed with them matches the norm that makes its own pass."""
    from sggan_amd import kernels as K, _abi as A
    N, H, W = 2, 32, 64
    rng = np.random.default_rng(9)
    x = torch.zeros((N, H, W, 8), dtype=torch.bfloat16, device="cuda")
    x[..., :3] = dev(rng.uniform(0, 1, (N, H, W, 3)), torch.bfloat16)
    w = dev(rng.standard_normal((7, 7, 3, 64)) / np.sqrt(49 * 3))
    b = dev(rng.standard_normal(64))
    g = K.conv_geom(N, H, W, 8, 64, 7, 7, 1, "VALID", 3, torch.bfloat16)
    assert g.stats_chunks == (H // 16) * (W // 32) and not g.pair_ok and not g.normload_ok
    wf, _ = K.pack_weights(w, 8, 64, torch.bfloat16)
    y0 = K.conv_fwd(g, x, wf, b)
    y1, part = K.conv_fwd_stats(g, x, wf, b)
    assert torch.equal(y0, y1)
    yf = y1.float().cpu().numpy().astype(np.float64)
    got = part.cpu().numpy().astype(np.float64).sum(axis=1)
    s1, s2 = yf.sum(axis=(1, 2)), (yf * yf).sum(axis=(1, 2))
    assert np.abs(got[..., 0] - s1).max() < 1e-4 * max(1.0, np.abs(s1).max()) + 1e-2
    assert np.abs(got[..., 1] - s2).max() < 1e-4 * s2.max()
    gam, bet = dev(1 + 0.2 * rng.standard_normal(64)), dev(0.2 * rng.standard_normal(64))
    za, sa = K.instnorm_fwd(y1, gam, bet, None, 1e-3, A.ACT_RELU)
    zb, sb = K.instnorm_fwd_partial(y1, part, gam, bet, None, 1e-3, A.ACT_RELU)
    assert (sa - sb).abs().max().item() < 1e-4 and (za.float() - zb.float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("pad,act", [("REFLECT-1", "relu"), ("SAME", None)], ids=["reflect_relu", "zero_none"])
def test_conv_dgrad_norm_bwd_epilogue(sg, pad, act):
    """conv data gradient + instance-norm-backward partial sums in its epilogue (sgg_conv2d_bwd_data_stats): dx is
    bit-identical to the plain data gradient and the norm backward fed with the partials matches the norm backward
    that makes its own statistics pass (dx, dgamma, dbeta)."""
    from sggan_amd import kernels as K, _abi as A
    N, H, W, Ci, Co = 2, 4, 128, 128, 192
    rng = np.random.default_rng(21)
    padding, refl = ("VALID", 1) if pad.startswith("REFLECT") else (pad, 0)
    g = K.conv_geom(N, H, W, Ci, Co, 3, 3, 1, padding, refl, torch.bfloat16)
    assert g.bwd_stats_chunks == (H // 2) * (W // 128) * 2
    w = dev(rng.standard_normal((3, 3, Ci, Co)) / np.sqrt(9 * Ci))
    _, wd = K.pack_weights(w, Ci, Co, torch.bfloat16)
    dy = dev(rng.standard_normal(g.y_shape), torch.bfloat16)
    addend = dev(rng.standard_normal(g.x_shape), torch.bfloat16)
    # the norm in front of the conv: its input nx, statistics, affine parameters
    nx = dev(rng.standard_normal(g.x_shape) * 1.3 + 0.2, torch.bfloat16)
    gam, bet = dev(1 + 0.2 * rng.standard_normal(Ci)), dev(0.2 * rng.standard_normal(Ci))
    a = {"relu": A.ACT_RELU, None: A.ACT_NONE}[act]
    _, st = K.instnorm_fwd(nx, gam, bet, None, 1e-3, a)
    dx0 = K.conv_dgrad(g, dy, wd, addend)
    dx1, part = K.conv_dgrad_stats(g, dy, wd, addend, nx, st, gam, bet, a, 0.0)
    assert torch.equal(dx0, dx1)
    dg0, db0, dg1, db1 = (torch.zeros(Ci, device="cuda") for _ in range(4))
    r0 = K.instnorm_bwd(dx0, nx, gam, bet, st, dg0, db0, False, a)
    r1 = K.instnorm_bwd_partial(dx1, nx, part, gam, bet, st, dg1, db1, False, a)
    sc = r0.float().abs().max().item()
    assert (r0.float() - r1.float()).abs().max().item() < 2e-2 * sc
    assert (dg0 - dg1).abs().max().item() < 1e-3 * max(1.0, dg0.abs().max().item())
    assert (db0 - db1).abs().max().item() < 1e-3 * max(1.0, db0.abs().max().item())


def test_conv_wgrad_pair(sg):
    """sgg_conv2d_bwd_weight_pair: one launch for two applications of a layer == the two separate weight gradients."""
    from sggan_amd import kernels as K
    N, H, W, Ci, Co = 2, 4, 128, 64, 128
    rng = np.random.default_rng(8)
    g = K.conv_geom(N, H, W, Ci, Co, 3, 3, 1, "VALID", 1, torch.bfloat16)
    assert g.wgrad_pair
    xs = [dev(rng.standard_normal(g.x_shape), torch.bfloat16) for _ in range(2)]
    dys = [dev(rng.standard_normal(g.y_shape), torch.bfloat16) for _ in range(2)]
    ref = torch.zeros((3, 3, Ci, Co), device="cuda")
    K.conv_wgrad(g, xs[0], dys[0], ref, accumulate=True)
    K.conv_wgrad(g, xs[1], dys[1], ref, accumulate=True)
    got = torch.full((3, 3, Ci, Co), 1.0, device="cuda")
    K.conv_wgrad_pair(g, xs[0], dys[0], xs[1], dys[1], got, accumulate=True)
    err = (got - 1.0 - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-5, err
    g2 = K.conv_geom(N, 5, 7, Ci, Co, 3, 3, 1, "VALID", 1, torch.bfloat16)
    assert not g2.wgrad_pair


DECONV_CASES = [("d1_like", 16, 8, 6, 5), ("d2_like", 24, 16, 4, 8), ("wide", 128, 64, 5, 7), ("odd_c", 8, 3, 3, 3),
                ("s2halo_d2_like", 128, 64, 8, 32), ("s2halo_d1_like", 256, 128, 16, 32),
                ("w9s2_d2_like", 128, 64, 4, 64)]


@pytest.mark.parametrize("dtype", DT, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", DECONV_CASES, ids=[c[0] for c in DECONV_CASES])
def test_deconv2d_fwd_bwd(sg, case, dtype):
    _, Ci, Co, H, W = case
    rng = np.random.default_rng(11)
    q = (lambda a: dev(a, dtype).detach().float().cpu().numpy().astype(np.float64))
    x = q(rng.standard_normal((2, H, W, Ci)))
    w = q(rng.standard_normal((3, 3, Co, Ci)) / np.sqrt(9 * Ci))
    b = rng.standard_normal(Co).astype(np.float32).astype(np.float64)
    t = O.Tape()
    vx, vw, vb = V(x), V(w), V(b)
    y = O.deconv2d(t, vx, vw, vb, 2)
    dy = q(rng.standard_normal(y.v.shape))
    t.backward([(y, dy)])
    tx, tw, tb = dev(x, dtype).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    ty = sg.deconv2d(tx, tw, tb, stride=2)
    assert tuple(ty.shape) == (2, 2 * H, 2 * W, Co)
    close(ty.detach().float().cpu().numpy(), y.v, dtype, "y")
    ty.backward(dev(dy, dtype))
    close(tx.grad.detach().float().cpu().numpy(), vx.g, dtype, "dx")
    close(tw.grad.detach().cpu().numpy(), vw.g, dtype, "dw")
    close(tb.grad.detach().cpu().numpy(), vb.g, dtype, "db")


@pytest.mark.parametrize("case", [("d2_like", 128, 64, 8, 32, 2), ("d1_like_two_channel_tiles", 256, 128, 16, 64, 1), ("pair", 128, 64, 16, 32, 4)],
                         ids=lambda c: c[0])
def test_deconv_fwd_stats_epilogue(sg, case):
    """Conv2DTranspose forward + instance-norm statistics epilogue (sgg_deconv2d_fwd_stats, module.py:254-260): the output is
    bit-identical to the plain forward (for a stacked pair: to the two networks' plain forwards), the per-chunk (sum, sumsq) rows
    -- one per 16 x 64 output-pixel tile -- add up to the sums of the STORED bf16 output, and the norm fed with them matches the norm
    that makes its own statistics pass."""
    from sggan_amd import kernels as K, _abi as A
    name, Ci, Co, H, W, N = case
    rng = np.random.default_rng(5)
    x = dev(rng.standard_normal((N, H, W, Ci)), torch.bfloat16)
    mk = lambda: (dev(rng.standard_normal((3, 3, Co, Ci)) / np.sqrt(9 * Ci)), dev(rng.standard_normal(Co)))
    (w, b), (w2, b2) = mk(), mk()
    g = K.deconv_geom(N, H, W, Ci, Co, 3, 3, 2, torch.bfloat16)
    assert g.stats_chunks == (H // 8) * (W // 32)
    _, wd = K.pack_weights(w, Co, Ci, torch.bfloat16)       # (kh,kw,out,in) IS the HWIO kernel of the equivalent conv: C = out, K = in
    _, wd2 = K.pack_weights(w2, Co, Ci, torch.bfloat16)
    if name == "pair":
        h = N // 2
        gh = K.deconv_geom(h, H, W, Ci, Co, 3, 3, 2, torch.bfloat16)
        y0 = torch.cat([K.deconv_fwd(gh, x[:h], wd, b), K.deconv_fwd(gh, x[h:], wd2, b2)])
        y1, part = K.deconv_fwd_stats(g, x, wd, b, pair=(wd2, b2, h))
    else:
        y0 = K.deconv_fwd(g, x, wd, b)
        y1, part = K.deconv_fwd_stats(g, x, wd, b)
    assert torch.equal(y0, y1)
    yf = y1.float().cpu().numpy().astype(np.float64)
    got = part.cpu().numpy().astype(np.float64).sum(axis=1)                   # (N, Co, 2)
    s1, s2 = yf.sum(axis=(1, 2)), (yf * yf).sum(axis=(1, 2))
    assert np.abs(got[..., 0] - s1).max() < 1e-4 * max(1.0, np.abs(s1).max()) + 1e-2
    assert np.abs(got[..., 1] - s2).max() < 1e-4 * s2.max()
    gam, bet = dev(1 + 0.2 * rng.standard_normal(Co)), dev(0.2 * rng.standard_normal(Co))
    za, sa = K.instnorm_fwd(y1, gam, bet, None, 1e-3, A.ACT_RELU)
    zb, sb = K.instnorm_fwd_partial(y1, part, gam, bet, None, 1e-3, A.ACT_RELU)
    assert (sa - sb).abs().max().item() < 1e-4
    assert (za.float() - zb.float()).abs().max().item() < 2e-2
    # shapes the stride-2 halo kernel does not take report 0 chunks
    assert K.deconv_geom(N, 5, 7, Ci, Co, 3, 3, 2, torch.bfloat16).stats_chunks == 0
    assert K.deconv_geom(N, H, W, Ci, Co, 3, 3, 2, torch.float32).stats_chunks == 0


@pytest.mark.parametrize("dtype", DT, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape,act", [((2, 9, 7, 16), None), ((2, 32, 40, 64), "relu"), ((3, 17, 19, 40), "lrelu"),
                                       ((2, 1, 1, 512), "lrelu"), ((1, 64, 64, 8), "relu"), ((2, 5, 13, 520), None)])
def test_instance_norm_fwd_bwd(sg, shape, act, dtype):
    rng = np.random.default_rng(5)
    q = (lambda a: dev(a, dtype).detach().float().cpu().numpy().astype(np.float64))
    C_ = shape[-1]
    x = q(rng.standard_normal(shape) * 1.5 + 0.3)
    g = (1 + 0.2 * rng.standard_normal(C_)).astype(np.float32).astype(np.float64)
    b = (0.2 * rng.standard_normal(C_)).astype(np.float32).astype(np.float64)
    t = O.Tape()
    vx, vg, vb = V(x), V(g), V(b)
    y = O.instance_norm(t, vx, vg, vb, 1e-3)
    if act == "relu":
        y = O.relu(t, y)
    elif act == "lrelu":
        y = O.lrelu(t, y, 0.3)
    dy = q(rng.standard_normal(shape))
    t.backward([(y, dy)])
    tx, tg, tb = dev(x, dtype).requires_grad_(True), dev(g).requires_grad_(True), dev(b).requires_grad_(True)
    ty = sg.instance_norm(tx, tg, tb, eps=1e-3, act=act, leak=0.3)
    close(ty.detach().float().cpu().numpy(), y.v, dtype, "y")
    ty.backward(dev(dy, dtype))
    close(tx.grad.detach().float().cpu().numpy(), vx.g, dtype, "dx", scale=max(np.abs(vx.g).max(), 1e-3))
    close(tg.grad.detach().cpu().numpy(), vg.g, dtype, "dgamma", scale=max(np.abs(vg.g).max(), 1.0))
    close(tb.grad.detach().cpu().numpy(), vb.g, dtype, "dbeta", scale=max(np.abs(vb.g).max(), 1.0))


def test_instance_norm_residual_and_eps(sg):
    from sggan_amd import kernels as K
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 8, 8, 16)); r = rng.standard_normal((2, 8, 8, 16))
    g = np.ones(16); b = np.zeros(16)
    for eps in (1e-3, 1e-5):                                      # live tfa eps and the ops.py:19 spec
        t = O.Tape()
        y = O.add(t, O.instance_norm(t, V(x), V(g), V(b), eps), V(r))
        ty, _ = K.instnorm_fwd(dev(x), dev(g), dev(b), dev(r), eps)
        close(ty.detach().cpu().numpy(), y.v, torch.float32, f"eps={eps}")


@pytest.mark.parametrize("dtype", DT, ids=["f32", "bf16"])
def test_activations(sg, dtype):
    rng = np.random.default_rng(4)
    x = dev(rng.standard_normal((2, 5, 7, 24)), dtype)
    xf = x.detach().float().cpu().numpy().astype(np.float64)
    for fn, ref, dref in ((lambda t: sg.lrelu(t, 0.3), lambda a: np.where(a > 0, a, 0.3 * a), lambda a: np.where(a > 0, 1, 0.3)),
                          (lambda t: sg.lrelu(t, 0.2), lambda a: np.maximum(a, 0.2 * a), lambda a: np.where(a > 0, 1, 0.2)),
                          (sg.relu, lambda a: np.maximum(a, 0), lambda a: (a > 0) * 1.0),
                          (sg.tanh, np.tanh, lambda a: 1 - np.tanh(a) ** 2)):
        tx = x.clone().requires_grad_(True)
        y = fn(tx)
        close(y.detach().float().cpu().numpy(), ref(xf), dtype, "act")
        y.backward(torch.ones_like(y))
        close(tx.grad.detach().float().cpu().numpy(), dref(xf), dtype, "dact")


def test_instnorm_bwd_mixed_f32_gradient_of_bf16_tensor(sg):
    """sgg_instnorm_bwd_mixed: dy in float32 against a bf16 tensor (mixed mode) vs the oracle on the same bf16-rounded x and the
    f32 dy; dgamma / dbeta (f32) at 1e-4, dx (stored bf16) at one bf16 rounding."""
    from sggan_amd import kernels as K, _abi as A
    rng = np.random.default_rng(12)
    for shape, act in (((2, 16, 64, 64), A.ACT_RELU), ((1, 40, 24, 256), A.ACT_NONE), ((2, 5, 7, 32), A.ACT_LRELU)):
        C = shape[-1]
        tx = dev(rng.standard_normal(shape) * 1.5 + 0.3, torch.bfloat16)
        xq = tx.float().cpu().numpy().astype(np.float64)
        dy = rng.standard_normal(shape).astype(np.float32)
        gam, bet = (1 + 0.2 * rng.standard_normal(C)).astype(np.float32), (0.2 * rng.standard_normal(C)).astype(np.float32)
        t = O.Tape()
        vx, vg, vb = V(xq), V(gam), V(bet)
        y = O.instance_norm(t, vx, vg, vb, 1e-3)
        if act == A.ACT_RELU:
            y = O.relu(t, y)
        elif act == A.ACT_LRELU:
            y = O.lrelu(t, y, 0.3)
        t.backward([(y, dy.astype(np.float64))])
        _, st = K.instnorm_fwd(tx, dev(gam), dev(bet), None, 1e-3, act, 0.3)
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        dx = K.instnorm_bwd(dev(dy), tx, dev(gam), dev(bet), st, dg, db, False, act, 0.3)
        assert dx.dtype == torch.bfloat16
        close(dg.cpu().numpy(), vg.g, torch.float32, "dgamma", scale=np.abs(vg.g).max())
        close(db.cpu().numpy(), vb.g, torch.float32, "dbeta", scale=max(np.abs(vb.g).max(), 1e-3 * np.abs(dy).sum() / C))
        err = np.abs(dx.float().cpu().numpy() - vx.g).max() / np.abs(vx.g).max()
        assert err < 6e-3, (shape, err)                      # one bf16 rounding of the result (2^-8 = 3.9e-3 of the value)


def test_seg_class_map_reference_fixtures_bit_exact(sg):
    sc = sg.segment_class
    z = np.load(os.path.join(G, "segclass_gta.npz"))
    for d in ("trainA", "trainB"):
        for key in ("crop_rgb", "crop_rgba"):                      # alpha ignored (segment_class.py:97 img[x,y,:3])
            got = sc.preprocess(z[f"{d}_{key}"]).detach().cpu().numpy()
            assert np.array_equal(got, z[f"{d}_crop_expected"]), (d, key)
    # every table entry, near misses, default
    cols = np.array([k for k, _ in O.CITYSCAPE_MAP] + [(0, 0, 0), (128, 64, 129), (129, 64, 128), (255, 255, 255)], np.uint8)
    exp = np.array([v for _, v in O.CITYSCAPE_MAP] + [0, 0, 0, 0], np.uint8)
    assert np.array_equal(sc.preprocess(cols[None]).detach().cpu().numpy()[0], exp)
    assert dict(sc.cityscape()) == {k: v for k, v in O.CITYSCAPE_MAP}
    # full Cityscapes-size image: random palette field, exact equality with the oracle
    rng = np.random.default_rng(0)
    pal = np.array([k for k, _ in O.CITYSCAPE_MAP] + [(0, 0, 0), (20, 20, 20), (111, 74, 0)], np.uint8)
    img = pal[rng.integers(0, len(pal), (1024, 2048))]
    got = sc.preprocess(img).detach().cpu().numpy()
    assert np.array_equal(got, O.seg_class_map(img))
    assert sc.preprocess(np.zeros((0, 4, 3), np.uint8)).shape == (0, 4)
    # ragged pixel counts (the 4-pixels-per-thread kernel + its scalar tail) and pointers that are not 4-byte aligned (scalar path)
    from sggan_amd import kernels as K
    for ch in (3, 4):
        for npx in (1, 2, 3, 4, 5, 7, 8, 1023):
            src = np.concatenate([pal[rng.integers(0, len(pal), npx)], np.full((npx, 1), 255, np.uint8)], 1)[:, :ch]
            for off in (0, 1, 2):
                buf = torch.zeros(off + npx * ch, dtype=torch.uint8, device="cuda")
                buf[off:] = torch.as_tensor(np.ascontiguousarray(src).ravel()).cuda()
                got = K.seg_class_map(buf[off:].view(npx, ch)).cpu().numpy()
                assert np.array_equal(got, O.seg_class_map(np.ascontiguousarray(src)[None, :, :])[0]), (ch, npx, off)


def test_onehot_resample_bit_exact(sg):
    sc = sg.segment_class
    rng = np.random.default_rng(1)
    for (H, W, oh, ow, nc) in ((128, 128, 4, 4, 34), (1024, 2048, 5, 13, 34), (256, 512, 8, 15, 8), (37, 53, 1, 1, 8), (16, 16, 16, 16, 8)):
        idx = rng.integers(0, nc, (2, H, W)).astype(np.uint8)
        got = sc.one_hot_mask(idx, oh, ow, nc).detach().cpu().numpy()
        exp = np.stack([O.mask_from_index(i, nc, oh, ow) for i in idx])
        assert np.array_equal(got, exp), (H, W, oh, ow)
        assert np.array_equal(got.sum(-1), np.ones((2, oh, ow)))


def test_onehot_resample_matches_reference_zoom_on_reference_maps(sg):
    """a12 pin on the GPU: sgg_onehot_resample on the reference's own class-index maps (datasets/city/trainA_seg_class)
    vs the committed outputs of the reference's call (one_hot + scipy.ndimage.zoom, utils.py:158-165,197-199)."""
    sc = sg.segment_class
    z = np.load(os.path.join(G, "mask_zoom_city.npz"))
    idx = np.stack([z[f"idx{i}"] for i in range(len(z["names"]))])
    for H, W in z["sizes"]:
        exp = np.stack([z[f"zoom{i}_{H}x{W}"] for i in range(len(idx))]).astype(np.float32)
        got = sc.one_hot_mask(idx, exp.shape[1], exp.shape[2], 34).detach().cpu().numpy()
        assert got.dtype == np.float32 and np.array_equal(got, exp), (H, W)


@pytest.mark.parametrize("dtype", DT, ids=["f32", "bf16"])
def test_mask_reduce(sg, dtype):
    from sggan_amd import kernels as K
    rng = np.random.default_rng(8)
    for hh, hw, mh, mw in ((1, 1, 4, 4), (5, 13, 5, 13)):
        h4 = rng.standard_normal((2, hh, hw, 34))
        h4p = np.pad(h4, ((0, 0),) * 3 + ((0, 6),))
        th = dev(h4p, dtype)
        h4q = th.detach().float().cpu().numpy()[..., :34].astype(np.float64)
        mask = np.stack([O.one_hot(i, 34) for i in rng.integers(0, 34, (2, mh, mw))]).astype(np.float64)
        t = O.Tape()
        vh = V(h4q)
        y = O.mask_reduce(t, vh, mask)
        dy = rng.standard_normal(y.v.shape)
        t.backward([(y, dy)])
        out = K.mask_reduce_fwd(th, dev(mask), 34)
        close(out.detach().cpu().numpy(), y.v, torch.float32, "mask fwd")
        dh = K.mask_reduce_bwd(dev(dy), dev(mask), tuple(th.shape), dtype, 34)
        close(dh.detach().float().cpu().numpy()[..., :34], vh.g, dtype, "mask bwd")
        assert float(dh.float()[..., 34:].abs().max()) == 0.0


def test_losses(sg):
    from sggan_amd import kernels as K
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 4, 4, 1)) * 3
    loss = torch.zeros(1, device="cuda")
    for label in (1.0, 0.0):
        t = O.Tape(); vx = V(x); l = O.bce_logits_mean(t, vx, label); t.backward([(l, 1.0)])
        dx = torch.empty(x.shape, device="cuda")
        K.bce_logits(dev(x), label, loss, dx)
        assert abs(loss.item() - l.v) < 1e-6 * max(1, abs(l.v))
        close(dx.detach().cpu().numpy(), vx.g, torch.float32, "dbce")
    K.bce_logits(dev(np.zeros(7)), 1.0, loss, None)
    assert abs(loss.item() - np.log(2)) < 1e-7
    K.bce_logits(dev(np.zeros(7)), 0.0, loss, None, accumulate_loss=True)
    assert abs(loss.item() - 2 * np.log(2)) < 1e-6
    for dtype in DT:
        a = np.zeros((2, 16, 16, 8)); b = np.zeros((2, 16, 16, 8))
        a[..., :3] = rng.uniform(0, 1, (2, 16, 16, 3)); b[..., :3] = np.tanh(rng.standard_normal((2, 16, 16, 3)))
        ta, tb = dev(a, dtype), dev(b, dtype)
        aq, bq = ta.detach().float().cpu().numpy().astype(np.float64)[..., :3], tb.detach().float().cpu().numpy().astype(np.float64)[..., :3]
        t = O.Tape(); vb = V(bq); l = O.l1_mean(t, aq, vb); t.backward([(l, 100.0)])
        db = torch.empty_like(tb)
        K.l1_loss(ta, tb, 3, loss, db, weight=100.0)
        assert abs(loss.item() - 100 * l.v) < 1e-4 * 100 * l.v
        close(db.detach().float().cpu().numpy()[..., :3], vb.g, dtype, "dl1")
        assert float(db.float()[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("eps", [1e-7, 1e-2])
def test_adam_tf_form(sg, eps):
    from sggan_amd import kernels as K
    rng = np.random.default_rng(3)
    n = 10007
    th, g = rng.standard_normal(n), rng.standard_normal(n) * 0.1
    m, v = np.zeros(n), np.zeros(n)
    tth, tm, tv = dev(th), dev(m), dev(v)
    for t in (1, 2, 3):
        g = rng.standard_normal(n) * 0.1
        th, m, v = O.adam_tf(th, g.astype(np.float32).astype(np.float64), m, v, t, 1e-3, 0.5, 0.999, eps)
        K.adam(tth, dev(g), tm, tv, t, 1e-3, 0.5, 0.999, eps)
    assert np.abs(tth.detach().cpu().numpy() - th).max() < 2e-6
    # step-1 known answer with g = 1: -lr*sqrt(1-b2)/(sqrt(1-b2)+eps)  (Keras form, not torch's)
    one = dev(np.zeros(4)); K.adam(one, dev(np.ones(4)), dev(np.zeros(4)), dev(np.zeros(4)), 1, 1e-3, 0.5, 0.999, eps)
    assert abs(one[0].item() + 1e-3 * np.sqrt(1e-3) / (np.sqrt(1e-3) + eps)) < 5e-9


def test_abi_rejects_bad_arguments(sg):
    from sggan_amd import kernels as K
    with pytest.raises(Exception):
        K.conv_geom(1, 4, 4, 8, 8, 3, 3, 1, "VALID", 5, torch.float32)       # reflect pad >= size
    with pytest.raises(Exception):
        K.conv_geom(1, 2, 2, 8, 8, 3, 3, 1, "VALID", 0, torch.float32)       # empty output
    with pytest.raises(sg.SggError):
        K.mask_reduce_fwd(torch.zeros(1, 2, 2, 40, device="cuda"), torch.zeros(1, 3, 3, 34, device="cuda"), 34)
    with pytest.raises(TypeError):
        K.act_fwd(torch.zeros(8, device="cuda", dtype=torch.float16), 1)


def test_cycle_criteria_mse_edge_gradloss(sg):
    """Defined-not-wired SG-GAN criteria: mae_criterion, seg-edge indicator, gradloss_criterion."""
    from sggan_amd import kernels as K
    rng = np.random.default_rng(12)
    loss = torch.zeros(1, device="cuda")
    x = rng.standard_normal((2, 5, 13, 1)) * 2
    for tgt in (1.0, 0.0):
        t = O.Tape(); vx = V(x); l = O.mse_const_mean(t, vx, tgt); t.backward([(l, 0.5)])
        dx = torch.empty(x.shape, device="cuda")
        K.mse_const(dev(x), tgt, loss, dx, weight=0.5)
        assert abs(loss.item() - 0.5 * l.v) < 1e-6 * max(1, l.v)
        close(dx.cpu().numpy(), vx.g, torch.float32, "dmse")
    # edge indicator: blocky colour map, exact {0,1} equality
    pal = rng.integers(0, 256, (6, 3)) / 255.0
    idx = np.repeat(np.repeat(rng.integers(0, 6, (2, 4, 6)), 8, 1), 8, 2)
    seg = pal[idx]
    for dtype in DT:
        got = K.seg_edge_weight(dev(np.pad(seg, ((0, 0),) * 3 + ((0, 5),)), dtype), 3).cpu().numpy()
        assert np.array_equal(got, O.seg_edge_weight(seg)[..., 0])
    # gradient-sensitive loss, forward and d/d(in)
    for dtype in DT:
        a = np.zeros((2, 16, 24, 8)); b = np.zeros((2, 16, 24, 8))
        a[..., :3] = np.tanh(rng.standard_normal((2, 16, 24, 3))); b[..., :3] = rng.uniform(0, 1, (2, 16, 24, 3))
        ta, tb = dev(a, dtype), dev(b, dtype)
        aq = ta.float().cpu().numpy().astype(np.float64)[..., :3]; bq = tb.float().cpu().numpy().astype(np.float64)[..., :3]
        w = (rng.uniform(size=(2, 16, 24, 1)) > 0.4) * 1.0
        t = O.Tape(); va = V(aq); l = O.gradloss(t, va, bq, w); t.backward([(l, 5.0)])
        da = torch.empty_like(ta)
        K.gradloss(ta, tb, dev(w[..., 0]), 3, loss, da, lam=5.0)
        assert abs(loss.item() - 5 * l.v) < (1e-5 if dtype == torch.float32 else 1e-3) * 5 * l.v
        close(da.float().cpu().numpy()[..., :3], va.g, dtype, "dgradloss")
        assert float(da.float()[..., 3:].abs().max()) == 0.0


def test_timed_launch_hooks_put_events_on_the_kernel_dispatch(sg):
    """bench.py's measurement hooks (sgg_time_next_launch): an armed event pair is consumed by the main kernel of the next
    timed-family launch and reads a plausible kernel duration; a call outside the timed families leaves it unused."""
    import ctypes
    from sggan_amd import kernels as K, _abi as A
    s, e = ctypes.c_void_p(), ctypes.c_void_p()
    A.check(A.lib().sgg_event_create(ctypes.byref(s)), "event_create")
    A.check(A.lib().sgg_event_create(ctypes.byref(e)), "event_create")
    g = K.conv_geom(2, 16, 128, 64, 64, 3, 3, 1, "VALID", 1, torch.bfloat16)            # a halo-GEMM shape
    x = torch.randn(g.x_shape, device="cuda").to(torch.bfloat16)
    wf, _ = K.pack_weights(torch.randn(3, 3, 64, 64, device="cuda") * 0.05, 64, 64, torch.bfloat16)
    K.conv_fwd(g, x, wf, None)                                                           # warm (LDS attribute, allocations)
    A.lib().sgg_time_next_launch(s, e)
    y = K.conv_fwd(g, x, wf, None)
    assert A.lib().sgg_time_next_launch(None, None) == 1                                 # consumed
    torch.cuda.synchronize()
    ms = ctypes.c_float()
    A.check(A.lib().sgg_event_elapsed_ms(s, e, ctypes.byref(ms)), "event_elapsed")
    assert 1e-3 < ms.value < 5.0, ms.value
    A.lib().sgg_time_next_launch(s, e)
    K.act_fwd(y, A.ACT_RELU)                                                             # not a timed family
    assert A.lib().sgg_time_next_launch(None, None) == 0
    y2 = K.conv_fwd(g, x, wf, None)                                                      # and the disarmed launch is an ordinary one
    assert torch.equal(y, y2)
    A.lib().sgg_event_destroy(s); A.lib().sgg_event_destroy(e)


@pytest.mark.parametrize("cfg", [
    # (N per network, H, W, C, K, R, stride, padding, dtype)
    ("D.h32-like: split-K tail", 2, 15, 31, 512, 512, 3, 2, "VALID", torch.bfloat16),
    ("D.h4-like: 34 output channels", 2, 5, 13, 512, 40, 3, 1, "SAME", torch.bfloat16),
    ("c2-like: stride-2 forward / stride-2 halo data gradient", 2, 64, 128, 64, 128, 3, 2, "SAME", torch.bfloat16),
    ("h0-like: 8 -> 64 channels, stride 2 (the narrow stride-2 data-gradient kernel)", 2, 32, 80, 8, 64, 3, 2, "SAME", torch.bfloat16),
    ("h31-like: 4x4 stride-2 VALID, 256 -> 256 (the LDS-DMA weight-gradient kernel, two networks in one launch)", 2, 18, 34, 256, 256, 4, 2, "VALID", torch.bfloat16),
    ("h3-like: 256x256 tiles", 2, 32, 64, 256, 512, 3, 1, "SAME", torch.bfloat16),
    ("odd generic f32", 3, 9, 11, 16, 24, 3, 1, "SAME", torch.float32),
    ("3x3 stride-1 halo shape, zero padding", 2, 64, 128, 64, 64, 3, 1, "SAME", torch.bfloat16),
], ids=lambda c: c[0].split(":")[0])
def test_group2_launches_equal_two_single_calls(sg, cfg):
    """sgg_conv2d_fwd_group2 / _bwd_data_group2 / sgg_deconv2d_*_group2: the same call site of two networks in one call must be
    BIT-IDENTICAL to two single calls (each group runs the single call's tiles and split-K), with and without the fused
    activation / bias / skip-gradient addend -- the lockstep pairs of the cycle step rely on it."""
    from sggan_amd import kernels as K
    from sggan_amd import _abi as A
    _, N, H, W, Ci, Co, R, stride, padding, dt = cfg
    g = K.conv_geom(N, H, W, Ci, Co, R, R, stride, padding, 0, dt)
    gen = torch.Generator().manual_seed(3)
    r = lambda *sh: torch.randn(sh, generator=gen).cuda()
    x = r(2 * N, H, W, Ci).to(dt)
    dy = r(*((2 * N,) + tuple(g.y_shape[1:]))).to(dt)
    add = r(2 * N, H, W, Ci).to(dt)
    ws_ = [K.pack_weights(r(R, R, Ci, Co) / (R * R * Ci) ** 0.5, Ci, Co, dt) for _ in range(2)]
    bs = [r(Co), r(Co)]
    for act in (A.ACT_NONE, A.ACT_LRELU):
        y2 = K.conv_fwd_group2(g, x, ws_[0][0], bs[0], ws_[1][0], bs[1], act, 0.3)
        for k in range(2):
            assert torch.equal(y2[k * N:(k + 1) * N], K.conv_fwd(g, x[k * N:(k + 1) * N], ws_[k][0], bs[k], act, 0.3)), ("fwd", act, k)
    for ad in (None, add):
        dx2 = K.conv_dgrad_group2(g, dy, ws_[0][1], ws_[1][1], ad)
        for k in range(2):
            ref = K.conv_dgrad(g, dy[k * N:(k + 1) * N], ws_[k][1], None if ad is None else ad[k * N:(k + 1) * N])
            assert torch.equal(dx2[k * N:(k + 1) * N], ref), ("dgrad", ad is not None, k)
    for accumulate in (False, True):                     # bias gradient of both networks in one call
        base = [r(Co), r(Co)]
        dbs = [b.clone() for b in base]
        K.bias_grad_group2(dy, dbs[0], dbs[1], accumulate=accumulate)
        for k in range(2):
            ref = base[k].clone()
            K.bias_grad(dy[k * N:(k + 1) * N], ref, accumulate=accumulate)
            assert torch.equal(dbs[k], ref), ("bias grad", accumulate, k)
    # weight gradients: two main kernels + ONE slab reduce vs two full calls (also accumulating onto existing values)
    # (the 3x3 halo weight-gradient kernels give each network half the split slabs in one launch: same sums in another f32 order)
    s2halo = stride in (1, 2) and R == 3 and Ci % 64 == 0 and Co % 128 == 0 and dt == torch.bfloat16   # (the all-taps halo kernels)
    s2halo = s2halo or (Co >= 256 and R * R * Ci >= 256)                                              # (and the LDS-DMA kernel)

    def same(a, b, what):
        if s2halo:
            rel = float((a - b).norm() / b.norm())
            assert rel < 1e-5, (what, rel)
        else:
            assert torch.equal(a, b), what

    for accumulate in (False, True):
        base = [r(R, R, Ci, Co), r(R, R, Ci, Co)]
        dws = [b.clone() for b in base]
        K.conv_wgrad_group2(g, x, dy, dws[0], dws[1], accumulate=accumulate)
        for k in range(2):
            ref = base[k].clone()
            K.conv_wgrad(g, x[k * N:(k + 1) * N], dy[k * N:(k + 1) * N], ref, accumulate=accumulate)
            same(dws[k], ref, ("wgrad", accumulate, k))
    if stride == 2 and padding == "SAME" and H % 2 == 0 and W % 2 == 0:
        # the same weights as a Conv2DTranspose (module.py:254,258): x' = (2N, H/2, W/2, Co) -> (2N, H, W, Ci)
        gd = K.deconv_geom(N, H // 2, W // 2, Co, Ci, R, R, 2, dt)
        xt = r(2 * N, H // 2, W // 2, Co).to(dt)
        bt = [r(Ci), r(Ci)]
        yt = K.deconv_fwd_group2(gd, xt, ws_[0][1], bt[0], ws_[1][1], bt[1], A.ACT_RELU, 0.0)
        dxt = K.deconv_dgrad_group2(gd, x, ws_[0][0], ws_[1][0])
        for k in range(2):
            sl = slice(k * N, (k + 1) * N)
            assert torch.equal(yt[sl], K.deconv_fwd(gd, xt[sl], ws_[k][1], bt[k], A.ACT_RELU, 0.0)), ("deconv fwd", k)
            assert torch.equal(dxt[sl], K.deconv_dgrad(gd, x[sl], ws_[k][0])), ("deconv dgrad", k)
        dwt = [torch.empty((R, R, Ci, Co), device="cuda") for _ in range(2)]          # Keras transpose kernel: (kh, kw, out, in)
        K.conv_wgrad_group2(gd, xt, x, dwt[0], dwt[1])
        for k in range(2):
            sl = slice(k * N, (k + 1) * N)
            ref = torch.empty_like(dwt[k])
            K.deconv_wgrad(gd, xt[sl], x[sl], ref)
            same(dwt[k], ref, ("deconv wgrad", k))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    ("bench: residual-block shape, 8 images", 8, 64, 128, 256, 256),
    ("small map, 4 row tiles", 3, 8, 128, 64, 64),
    ("wide map, 128 source channels", 2, 16, 256, 128, 64),
    ("512 source channels", 2, 6, 128, 512, 256),
], ids=lambda c: c[0].split(":")[0])
def test_normalise_on_load_conv_equals_norm_then_conv(sg, cfg):
    """sgg_conv2d_fwd_stats_normload -- the residual block's conv -> InstanceNormalization -> ReLU -> conv (module.py:211-215)
    with the norm applied to the operand tiles inside the second conv -- must be BIT-IDENTICAL to the norm's apply pass
    followed by the plain conv: the normalised tensor it writes on the way, the conv output and the statistics rows; one
    network and the lockstep pair."""
    from sggan_amd import kernels as K
    from sggan_amd import _abi as A
    _, N, H, W, Ci, Co = cfg
    dt = torch.bfloat16
    g = K.conv_geom(N, H, W, Ci, Co, 3, 3, 1, "VALID", 1, dt)              # REFLECT pad 1
    assert g.normload_ok and g.stats_chunks > 0
    gen = torch.Generator().manual_seed(11)
    r = lambda *sh: torch.randn(sh, generator=gen).cuda()
    x_raw = (r(N, H, W, Ci) * 1.7 + 0.3).to(dt)
    # statistics rows of x_raw in the producing conv's format: any chunking works for the finalize step
    xf = x_raw.float().reshape(N, 4, H * W // 4, Ci)
    part = torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=-1).contiguous()
    gam, bet = [r(Ci) * 0.5 + 1.0 for _ in range(2)], [r(Ci) * 0.3 for _ in range(2)]
    ws_ = [K.pack_weights(r(3, 3, Ci, Co) / (9 * Ci) ** 0.5, Ci, Co, dt) for _ in range(2)]
    bs = [r(Co), r(Co)]
    # reference: finalize + apply (ReLU), then the conv with its statistics epilogue
    a_ref, st_ref = K.instnorm_fwd_partial(x_raw, part, gam[0], bet[0], None, 1e-3, A.ACT_RELU, 0.0)
    y_ref, p_ref = K.conv_fwd_stats(g, a_ref, ws_[0][0], bs[0])
    stats = K.instnorm_finalize(part, H * W, 1e-3)
    assert torch.equal(stats, st_ref)
    a1, y, p = K.conv_fwd_stats_normload(g, x_raw, stats, gam[0], bet[0], ws_[0][0], bs[0])
    assert torch.equal(a1, a_ref), "normalised operand"
    assert torch.equal(y, y_ref), "conv output"
    assert torch.equal(p, p_ref), "statistics rows"
    if N >= 2 and g.pair_ok:
        ns = N // 2
        a_ref2, _ = K.instnorm_fwd_partial_pair(x_raw, part, gam[0], bet[0], gam[1], bet[1], ns, None, 1e-3, A.ACT_RELU, 0.0)
        y_ref2, p_ref2 = K.conv_fwd_stats_pair(g, a_ref2, ws_[0][0], bs[0], ws_[1][0], bs[1], ns)
        a2, y2, p2 = K.conv_fwd_stats_normload(g, x_raw, stats, gam[0], bet[0], ws_[0][0], bs[0], pair=(gam[1], bet[1], ws_[1][0], bs[1], ns))
        assert torch.equal(a2, a_ref2) and torch.equal(y2, y_ref2) and torch.equal(p2, p_ref2), "pair"


@pytest.mark.gpu
def test_halo_kernels_with_operands_in_the_upper_half_of_a_4gib_window(sg):
    """The 3x3 halo kernels address their LDS-DMA sources as a uniform 64-bit base (SGPR pair, assembled from two
    v_readfirstlane halves) + a 32-bit lane offset.  Regression test for the sign extension of the low half: operands whose
    address has bit 31 set (the upper 2 GiB of every 4 GiB window) must give the same results as anywhere else."""
    from sggan_amd import kernels as K
    N, H, W, Cc = 2, 8, 128, 64
    dt = torch.bfloat16
    g = K.conv_geom(N, H, W, Cc, Cc, 3, 3, 1, "VALID", 1, dt)
    gen = torch.Generator().manual_seed(5)
    r = lambda *sh: torch.randn(sh, generator=gen).cuda()
    x, dy = r(N, H, W, Cc).to(dt), r(N, H, W, Cc).to(dt)
    wf, wd = K.pack_weights(r(3, 3, Cc, Cc) / 24.0, Cc, Cc, dt)
    bias = r(Cc)
    y_ref, dx_ref = K.conv_fwd(g, x, wf, bias), K.conv_dgrad(g, dy, wd)
    arena = torch.empty((1 << 32) + (1 << 28), dtype=torch.uint8, device="cuda")      # spans a whole 4 GiB window
    base = arena.data_ptr()
    first = (base >> 32 << 32) + 0x80001000                                            # low 32 address bits = 0x80001000
    off = (first if first >= base else first + (1 << 32)) - base
    assert 0 <= off and off + (64 << 20) < arena.numel()

    def place(t):
        nonlocal off
        nb = t.numel() * t.element_size()
        v = arena[off:off + nb].view(t.dtype).reshape(t.shape)
        v.copy_(t)
        assert (v.data_ptr() & 0xffffffff) >= (1 << 31)
        off += (nb + 255) // 256 * 256
        return v

    xh, dyh, wfh, wdh = place(x), place(dy), place(wf), place(wd)
    assert torch.equal(K.conv_fwd(g, xh, wfh, bias), y_ref)
    assert torch.equal(K.conv_dgrad(g, dyh, wdh), dx_ref)
    # the normalise-on-load variant is the one kernel that uses the SGPR-pair form
    xf = x.float().reshape(N, 2, H * W // 2, Cc)
    stats = K.instnorm_finalize(torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=-1).contiguous(), H * W, 1e-3)
    gam, bet = r(Cc) * 0.5 + 1.0, r(Cc) * 0.3
    ref = K.conv_fwd_stats_normload(g, x, stats, gam, bet, wf, bias)
    got = K.conv_fwd_stats_normload(g, xh, stats, gam, bet, wfh, bias)
    assert all(torch.equal(u, v) for u, v in zip(got, ref))
