"""Exact-arithmetic parity of the timed convolution kernels at the REAL layer shapes of the bench workload
(BASELINE configs[2]: 512x256, batch 8).

Inputs, weights, biases and upstream gradients are small integers: exactly representable in bf16 and f32, every product
an integer, every partial sum far below 2^24, so the f32 MFMA accumulation is exact in ANY order.  The float64 oracle
(oracle.sggan_oracle.conv2d / deconv2d, im2col GEMM) is then exact too and the kernel result must equal it BIT FOR BIT:
weight / bias gradients (f32) as they stand, activations and data gradients after the one rounding to the storage dtype.
This pins the bf16-only fast paths -- the LDS-resident halo GEMMs (forward, REFLECT data gradient with the mirror fold,
skip-gradient addend, norm-statistics epilogue), the all-taps weight-gradient kernels (plain, paired, stride 2), the
stride-2 parity-class data gradient / transposed conv, the 7x7 stem / head kernels and D's split-K tail -- which the
tolerance tests in test_gpu_ops.py only see at small sizes.
"""
import zlib

import numpy as np
import pytest
import torch

from oracle import sggan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sg():
    import sggan_amd
    return sggan_amd


def ints(rng, shape, lo=-1, hi=1):
    return rng.integers(lo, hi + 1, shape).astype(np.float64)


def store(a, dtype):
    """What an exact real-valued result looks like after ONE round-to-nearest-even to the storage dtype."""
    return torch.as_tensor(np.asarray(a, np.float64)).to(torch.float32).to(dtype).to(torch.float64).numpy()


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a, np.float32)).to("cuda").to(dtype)


def same(got, exp, what):
    got = got.detach().to(torch.float64).cpu().numpy()
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    bad = got != exp
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} elements differ, first at {np.argwhere(bad)[0].tolist()} " \
                          f"(got {got[bad][0]}, expected {exp[bad][0]}); max |diff| {np.abs(got - exp).max()}"


# (name, kind, R, stride, padding, Cin, Cout, N, H, W) -- the layers of generator_resnet / discriminator at 256x512, batch 8
LAYERS = [
    ("G.res_3x3_256", "conv", 3, 1, "REFLECT-1", 256, 256, 8, 64, 128),     # module.py:210-216 (18 per generator)
    ("G.c2_s2_64_128", "conv", 3, 2, "SAME", 64, 128, 8, 256, 512),         # :236
    ("G.c3_s2_128_256", "conv", 3, 2, "SAME", 128, 256, 8, 128, 256),       # :240
    ("G.d1_T_256_128", "deconv", 3, 2, "SAME", 256, 128, 8, 64, 128),       # :254
    ("G.d2_T_128_64", "deconv", 3, 2, "SAME", 128, 64, 8, 128, 256),        # :258
    ("G.c1_7x7_3_64", "conv", 7, 1, "REFLECT-3", 3, 64, 8, 256, 512),       # :230-232
    ("G.out_7x7_64_3", "conv", 7, 1, "REFLECT-3", 64, 3, 8, 256, 512),      # :262-264 (oracle evaluated 2 images at a time: its im2col is 3.3 GB per image)
    ("D.h0_s2_3_64", "conv", 3, 2, "SAME", 3, 64, 8, 256, 512),             # :284
    ("D.h1_s2_64_128", "conv", 3, 2, "SAME", 64, 128, 8, 128, 256),         # :287
    ("D.h2_s2_128_256", "conv", 3, 2, "SAME", 128, 256, 8, 64, 128),        # :291
    ("D.h3_s1_256_512", "conv", 3, 1, "SAME", 256, 512, 8, 32, 64),         # :295
    ("D.h31_s2v_512", "conv", 3, 2, "VALID", 512, 512, 8, 32, 64),          # :299
    ("D.h32_s2v_512", "conv", 3, 2, "VALID", 512, 512, 8, 15, 31),          # :303
    ("D.h33_s1v_512", "conv", 3, 1, "VALID", 512, 512, 8, 7, 15),           # :307
    ("D.h4_s1_512_34", "conv", 3, 1, "SAME", 512, 34, 8, 5, 13),            # :311
]


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_layer_fwd_bwd_bit_exact_at_bench_shape(sg, layer):
    name, kind, R, stride, padding, Ci, Co, N, H, W = layer
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = ints(rng, (N, H, W, Ci))
    b = ints(rng, (Co,), -2, 2)
    w = ints(rng, (R, R, Ci, Co) if kind == "conv" else (R, R, Co, Ci))       # Keras Conv2DTranspose kernel: (kh,kw,out,in)
    pad, refl = ("VALID", int(padding.split("-")[1])) if padding.startswith("REFLECT") else (padding, 0)
    # the oracle is per image (a convolution has no cross-image term; dw / db are sums over images, exact in integers), so a
    # layer whose im2col does not fit is evaluated in image slices -- the kernel still runs the whole batch in one launch
    chunk = 2 if name == "G.out_7x7_64_3" else N
    ys, dxs, dys, dw, db = [], [], [], 0.0, 0.0
    for n0 in range(0, N, chunk):
        t = O.Tape()
        vx, vb, vw = O.Var(x[n0:n0 + chunk]), O.Var(b), O.Var(w)
        yk = O.conv2d(t, vx, vw, vb, stride, pad, refl) if kind == "conv" else O.deconv2d(t, vx, vw, vb, stride)
        dyk = ints(rng, yk.v.shape)
        t.backward([(yk, dyk)])
        ys.append(yk.v); dxs.append(vx.g); dys.append(dyk)
        dw, db = dw + vw.g, db + vb.g
        del t, vx, yk
    from types import SimpleNamespace as NS
    y, vx, vw, vb, dy = NS(v=np.concatenate(ys)), NS(g=np.concatenate(dxs)), NS(g=dw), NS(g=db), np.concatenate(dys)
    del ys, dxs, dys
    assert max(np.abs(y.v).max(), np.abs(vx.g).max(), np.abs(vw.g).max(), np.abs(vb.g).max()) < 2 ** 24   # exactness precondition
    assert np.abs(y.v).max() > 8 and np.abs(vw.g).max() > 8                                              # and not trivial
    for dtype in (torch.bfloat16, torch.float32):
        tx = dev(x, dtype).requires_grad_(True)
        tw = dev(w).requires_grad_(True)
        tb = dev(b).requires_grad_(True)
        ty = sg.conv2d(tx, tw, tb, stride=stride, padding=padding) if kind == "conv" else sg.deconv2d(tx, tw, tb, stride=stride)
        tag = f"{name}[{str(dtype).split('.')[-1]}]"
        same(ty, store(y.v, dtype), tag + " y")
        ty.backward(dev(dy, dtype))
        same(tx.grad, store(vx.g, dtype), tag + " dx")
        same(tw.grad, vw.g, tag + " dw")
        same(tb.grad, vb.g, tag + " db")
        del tx, tw, tb, ty


def test_residual_conv_fused_variants_bit_exact_at_bench_shape(sg):
    """The residual-block conv's fused launches at N=8, 64x128, 256->256, REFLECT 1 (the roofline kernel of bench.py):
    forward + norm-statistics epilogue, data gradient + skip-gradient addend, paired weight gradient (two applications
    of the layer in one launch, the cycle step's form)."""
    from sggan_amd import kernels as K
    N, H, W, C = 8, 64, 128, 256
    rng = np.random.default_rng(77)
    x0, x1 = ints(rng, (N, H, W, C)), ints(rng, (N, H, W, C))
    w, b = ints(rng, (3, 3, C, C)), ints(rng, (C,), -2, 2)
    dy0, dy1 = ints(rng, (N, H, W, C)), ints(rng, (N, H, W, C))
    add = ints(rng, (N, H, W, C), -3, 3)
    res = []
    for x, dy in ((x0, dy0), (x1, dy1)):
        t = O.Tape()
        vx, vw, vb = O.Var(x), O.Var(w), O.Var(b)
        y = O.conv2d(t, vx, vw, vb, 1, "VALID", 1)
        t.backward([(y, dy)])
        res.append((y.v, vx.g, vw.g))
    bf = torch.bfloat16
    g = K.conv_geom(N, H, W, C, C, 3, 3, 1, "VALID", 1, bf)
    assert g.stats_chunks > 0 and g.wgrad_pair
    wf, wd = K.pack_weights(dev(w), C, C, bf)
    # forward with the statistics epilogue: same output bits; (sum, sumsq) rows = exact sums of the STORED bf16 output
    yq = store(res[0][0], bf)
    y1, part = K.conv_fwd_stats(g, dev(x0, bf), wf, dev(b))
    same(y1, yq, "fwd_stats y")
    got = part.to(torch.float64).sum(dim=1).cpu().numpy()                    # (N, C, 2): chunks combined
    assert np.abs(yq).max() ** 2 * 128 < 2 ** 24                              # every per-chunk sum is an exact f32 integer
    assert np.array_equal(got[..., 0], yq.sum(axis=(1, 2))) and np.array_equal(got[..., 1], (yq * yq).sum(axis=(1, 2)))
    # data gradient with the skip-connection gradient added in the epilogue (module.py:217 `y + x`)
    dx = K.conv_dgrad(g, dev(dy0, bf), wd, dev(add, bf))
    same(dx, store(res[0][1] + add, bf), "dgrad + addend")
    # two applications' weight gradients in one launch, accumulated onto an integer-valued dw
    dw = torch.full((3, 3, C, C), 5.0, device="cuda")
    K.conv_wgrad_pair(g, dev(x0, bf), dev(dy0, bf), dev(x1, bf), dev(dy1, bf), dw, accumulate=True)
    same(dw, res[0][2] + res[1][2] + 5.0, "wgrad pair")


def test_mixed_precision_data_gradient_is_exact_f32_at_bench_shape(sg):
    """Mixed mode (sgg_conv2d_bwd_data_mixed): bf16 operands, F32 result.  With integer inputs the result must equal the
    oracle's exact integers WITHOUT any rounding -- the REFLECT fold and a float32 skip-gradient addend included."""
    from sggan_amd import kernels as K
    N, H, W, C = 8, 64, 128, 256
    rng = np.random.default_rng(78)
    x, w, b = ints(rng, (N, H, W, C)), ints(rng, (3, 3, C, C)), ints(rng, (C,), -2, 2)
    dy = ints(rng, (N, H, W, C), -4, 4)
    add = ints(rng, (N, H, W, C), -1000, 1000) + 0.5          # not representable in bf16: the addend really is read as f32
    t = O.Tape()
    vx, vw, vb = O.Var(x), O.Var(w), O.Var(b)
    y = O.conv2d(t, vx, vw, vb, 1, "VALID", 1)
    t.backward([(y, dy)])
    assert np.abs(vx.g).max() > 256                            # above bf16's exact-integer range: a bf16 store would round
    g = K.conv_geom(N, H, W, C, C, 3, 3, 1, "VALID", 1, torch.bfloat16)
    assert g.dgrad_mixed
    _, wd = K.pack_weights(dev(w), C, C, torch.bfloat16)
    dx = K.conv_dgrad(g, dev(dy, torch.bfloat16), wd, None, out_f32=True)
    assert dx.dtype == torch.float32
    same(dx, vx.g, "mixed dgrad")
    dx2 = K.conv_dgrad(g, dev(dy, torch.bfloat16), wd, dev(add), out_f32=True)
    same(dx2, vx.g + add, "mixed dgrad + f32 addend")
    dx3 = K.conv_dgrad(g, dev(dy, torch.bfloat16), wd, dev(np.round(add / 8), torch.bfloat16), out_f32=True)
    same(dx3, vx.g + np.round(add / 8), "mixed dgrad + bf16 addend")


def test_two_network_weight_gradient_launch_bit_exact_at_bench_shape(sg):
    """sgg_conv2d_bwd_weight_pair2: both generators' weight gradients of a residual conv -- four (x, dy) sets of 8 x 64 x 128 x
    256, two dW -- in one launch, each network on half the blocks.  Integer inputs: both dW must equal the oracle exactly."""
    from sggan_amd import kernels as K
    N, H, W, C = 8, 64, 128, 256
    rng = np.random.default_rng(79)
    w, b = ints(rng, (3, 3, C, C)), np.zeros(C)
    sets, exp = [], []
    for _ in range(4):
        x, dy = ints(rng, (N, H, W, C)), ints(rng, (N, H, W, C))
        t = O.Tape()
        vx, vw, vb = O.Var(x), O.Var(w), O.Var(b)
        y = O.conv2d(t, vx, vw, vb, 1, "VALID", 1)
        t.backward([(y, dy)])
        sets.append((dev(x, torch.bfloat16), dev(dy, torch.bfloat16)))
        exp.append(vw.g)
    g = K.conv_geom(N, H, W, C, C, 3, 3, 1, "VALID", 1, torch.bfloat16)
    assert g.wgrad_pair
    dwa = torch.full((3, 3, C, C), 3.0, device="cuda")
    dwb = torch.full((3, 3, C, C), -7.0, device="cuda")
    ok = K.conv_wgrad_pair2(g, (*sets[0], *sets[1], dwa), (*sets[2], *sets[3], dwb), accumulate=True)
    assert ok
    same(dwa, exp[0] + exp[1] + 3.0, "network a")
    same(dwb, exp[2] + exp[3] - 7.0, "network b")
