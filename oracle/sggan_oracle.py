"""CPU oracle for the SG-GAN train-step hot path  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``sg-gan-tf2_amd/``) must never route through it.

It is a NumPy float64 restatement, written from the reference's source text, of
what ``fhfonsecaa/SG-GAN-TF2`` computes on the path ``BASELINE.json`` names.
Every function cites the reference ``file:line`` it follows (paths relative to
the reference root).

PARITY PINNING STATUS
---------------------
* ``seg_class_map`` (integer):  PINNED.  Checked bit-exact against the two
  input/output pairs the reference ships under ``datasets/gta/*_seg`` ->
  ``*_seg_class`` (crops + whole-image class histograms committed under
  ``tests/golden/``; generator: ``tests/golden/make_golden.py``).
* every floating-point function:  **PARITY UNPINNED**.  The arithmetic lives in
  third-party packages that are neither in the reference tree nor installed in
  the authoring container (tensorflow==2.1.0, tensorflow-addons==0.9.1,
  scipy==1.4.1, scikit-image==0.16.2; ``requirements_VP_project.txt:82-90``);
  the reference has no tests or golden vectors for them.  The restatement
  follows the published semantics of those packages (items tagged [3P] below),
  is cross-checked against an independent PyTorch-CPU composition
  (``oracle/torch_restatement.py``) and against hand-derived known answers
  (``tests/test_oracle_kat.py``).

A tiny reverse-mode tape gives gradients, so the whole ``train_step``
(``model.py:169-200``) -- forward, both losses, both gradient sets, both Adam
updates -- can be evaluated in float64.
"""
from __future__ import annotations

import numpy as np

F64 = np.float64

# ----------------------------------------------------------------------------
# integer path: colour -> class index, one-hot, mask resample
# ----------------------------------------------------------------------------

# segment_class.py:63-66 -- the 21 (R,G,B) -> class entries; everything else -> 0
# (``defaultdict(int)``, segment_class.py:61).
CITYSCAPE_MAP = (
    ((128, 64, 128), 4), ((244, 35, 232), 4), ((250, 170, 160), 4), ((230, 150, 140), 4),
    ((70, 70, 70), 5), ((102, 102, 156), 5), ((190, 153, 153), 5), ((180, 165, 180), 5),
    ((150, 100, 100), 5), ((150, 120, 90), 5), ((107, 142, 35), 7), ((70, 130, 180), 6),
    ((220, 20, 60), 2), ((255, 0, 0), 2), ((0, 0, 142), 1), ((0, 0, 70), 1),
    ((0, 60, 100), 1), ((0, 0, 90), 1), ((0, 0, 110), 1), ((0, 0, 230), 3), ((119, 11, 32), 3),
)


def seg_class_map(img: np.ndarray) -> np.ndarray:
    """segment_class.py:87-97 ``preprocess``: per-pixel ``maskmap[tuple(img[x,y,:3])]``.

    img: uint8 (M,N,3|4); alpha is ignored (``img[x,y,:3]``).  Returns uint8 (M,N).
    """
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] >= 3
    key = (img[..., 0].astype(np.uint32) << 16) | (img[..., 1].astype(np.uint32) << 8) | img[..., 2]
    out = np.zeros(img.shape[:2], np.uint8)
    for (r, g, b), v in CITYSCAPE_MAP:
        out[key == ((r << 16) | (g << 8) | b)] = v
    return out


def seg_class_map_loop(img: np.ndarray) -> np.ndarray:
    """Literal double loop of segment_class.py:95-97 (small inputs only)."""
    table = {k: v for k, v in CITYSCAPE_MAP}
    M, N = img.shape[:2]
    out = np.zeros((M, N), np.uint8)
    for x in range(M):
        for y in range(N):
            out[x, y] = table.get(tuple(int(t) for t in img[x, y, :3]), 0)
    return out


def one_hot(idx: np.ndarray, num_classes: int) -> np.ndarray:
    """utils.py:158-165 ``one_hot``: (M,N) int -> (M,N,C) int, hot[i,j,idx[i,j]] = 1."""
    idx = np.asarray(idx).astype(np.int64)
    hot = np.zeros(idx.shape + (num_classes,), np.int64)
    ii, jj = np.meshgrid(np.arange(idx.shape[0]), np.arange(idx.shape[1]), indexing="ij")
    hot[ii, jj, idx] = 1
    return hot


def zoom_mask_reference(hot: np.ndarray, image_height: int, image_width: int) -> np.ndarray:
    """utils.py:197-199: ``scipy.ndimage.zoom(one_hot, (H/34/h0, W/34/w0, 1), mode='nearest')``.

    [3P] scipy default order=3 (cubic spline), integer input -> integer output
    (rounded).  Output grid = round(H/34) x round(W/34).  Runs the scipy that is
    installed (1.15.x here; the reference pinned 1.4.1 -- the spline boundary
    handling for mode='nearest' changed in 1.6).
    """
    import scipy.ndimage
    z = (image_height / 34.0 / hot.shape[0], image_width / 34.0 / hot.shape[1], 1)
    return scipy.ndimage.zoom(hot, z, mode="nearest")


def resample_index_nearest(idx: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Align-corners nearest resample of a class-index map to (out_h,out_w).

    This is the coordinate rule scipy.ndimage.zoom uses (output sample i reads
    input coordinate i*(in-1)/(out-1)) at spline order 0 (round half to even is
    what ``np.rint`` does; scipy order-0 rounds half up: floor(x+0.5)).  It is the
    documented deviation D1/a12: the build's on-GPU mask pipeline
    resamples the *index* map then one-hots it, which equals one-hot-then-zoom
    whenever the cubic spline of a 0/1 field rounds to the nearest sample.
    """
    idx = np.asarray(idx)
    H, W = idx.shape

    def coords(n_in, n_out):
        if n_out == 1:
            return np.zeros(1, np.int64)
        c = np.arange(n_out, dtype=F64) * (n_in - 1) / (n_out - 1)
        return np.floor(c + 0.5).astype(np.int64)

    return idx[np.ix_(coords(H, out_h), coords(W, out_w))]


def mask_from_index(idx: np.ndarray, num_classes: int, out_h: int, out_w: int) -> np.ndarray:
    """one_hot(resample_index_nearest(idx)) as float64 (h,w,C) -- what D consumes."""
    return one_hot(resample_index_nearest(idx, out_h, out_w), num_classes).astype(F64)


# ----------------------------------------------------------------------------
# a minimal reverse-mode tape (float64)
# ----------------------------------------------------------------------------

class Var:
    __slots__ = ("v", "g", "name")

    def __init__(self, v, name=None):
        self.v = np.asarray(v, F64)
        self.g = None
        self.name = name

    def acc(self, g):
        self.g = g if self.g is None else self.g + g


class Tape:
    def __init__(self):
        self.ops = []

    def record(self, outs, fn):
        self.ops.append((outs, fn))

    def backward(self, seeds):
        """seeds: list of (Var, grad).  Runs every recorded vjp in reverse."""
        for v, g in seeds:
            v.acc(np.asarray(g, F64))
        for out, fn in reversed(self.ops):
            if out.g is not None:
                fn(out.g)


# ----------------------------------------------------------------------------
# primitives (forward value + vjp), NHWC
# ----------------------------------------------------------------------------

def reflect_index(n: int, p: int) -> np.ndarray:
    """[3P] tf.pad mode='REFLECT' (module.py:210,214,230,262): mirror WITHOUT
    repeating the edge sample.  Returns source indices for the padded axis."""
    assert p < n, "REFLECT pad needs pad < size"
    j = np.arange(-p, n + p)
    j = np.abs(j)
    j = np.where(j >= n, 2 * (n - 1) - j, j)
    return j


def same_pads(n_in: int, k: int, s: int):
    """[3P] TF 'SAME' padding: out = ceil(in/s); total = max((out-1)*s + k - in, 0);
    before = total // 2, after = total - before (the extra goes bottom/right)."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2, total - total // 2, out


def _im2col(xp, R, S, stride):
    win = np.lib.stride_tricks.sliding_window_view(xp, (R, S), axis=(1, 2))  # N,Ho',Wo',C,R,S
    win = win[:, ::stride, ::stride]
    N, Ho, Wo, C = win.shape[:4]
    cols = win.transpose(0, 1, 2, 4, 5, 3).reshape(N * Ho * Wo, R * S * C)
    return np.ascontiguousarray(cols), (N, Ho, Wo)


def conv2d(tape, x: Var, w: Var, b: Var, stride=1, padding="VALID", reflect=0) -> Var:
    """tf.keras.layers.Conv2D (module.py:211,215,232,236,240,264,284-311).

    w: HWIO (R,S,Cin,Cout) [3P Keras kernel layout]; b: (Cout,) (use_bias=True
    default [3P]).  padding: 'VALID' | 'SAME' ([3P] asymmetric, see same_pads).
    reflect=p first applies tf.pad(...,'REFLECT') of p on H and W
    (module.py:210,214,230,262) -- then padding must be 'VALID'.
    y[n,ho,wo,k] = b[k] + sum_{r,s,c} xpad[n, ho*stride + r, wo*stride + s, c] * w[r,s,c,k]
    """
    xv = x.v
    N, H, W, C = xv.shape
    R, S, Cw, K = w.v.shape
    assert Cw == C
    if reflect:
        assert padding == "VALID"
        ih, iw = reflect_index(H, reflect), reflect_index(W, reflect)
        xp = xv[:, ih][:, :, iw]
        pt = pl = None
    elif padding == "SAME":
        pt, pb, _ = same_pads(H, R, stride)
        pl, pr, _ = same_pads(W, S, stride)
        xp = np.pad(xv, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    else:
        pt = pl = 0
        xp = xv
    cols, (N_, Ho, Wo) = _im2col(xp, R, S, stride)
    wm = w.v.reshape(R * S * C, K)
    y = Var((cols @ wm + b.v).reshape(N, Ho, Wo, K))

    def vjp(gy):
        gm = gy.reshape(-1, K)
        w.acc((cols.T @ gm).reshape(w.v.shape))
        b.acc(gm.sum(0))
        gcols = (gm @ wm.T).reshape(N, Ho, Wo, R, S, C)
        gxp = np.zeros_like(xp)
        for r in range(R):
            for s in range(S):
                gxp[:, r:r + (Ho - 1) * stride + 1:stride, s:s + (Wo - 1) * stride + 1:stride] += gcols[:, :, :, r, s]
        if reflect:
            # MirrorPadGrad: fold the padded-grid gradient back, rows then columns
            tmp = np.zeros((N, H, xp.shape[2], C))
            for j, src in enumerate(ih):
                tmp[:, src] += gxp[:, j]
            gx = np.zeros_like(xv)
            for j, src in enumerate(iw):
                gx[:, :, src] += tmp[:, :, j]
        else:
            gx = gxp[:, pt:pt + H, pl:pl + W]
        x.acc(gx)

    tape.record(y, vjp)
    return y


def deconv2d(tape, x: Var, w: Var, b: Var, stride=2) -> Var:
    """tf.keras.layers.Conv2DTranspose(k, (3,3), strides=(2,2), padding='same')
    (module.py:254,258).  w: (R,S,Cout,Cin) [3P Keras transpose-kernel layout].

    [3P] Defined as the input-gradient of Conv2D(SAME, stride) whose input is the
    2H x 2W output: that conv pads (before, after) = same_pads(2H, R, stride), so
        full[n, i*stride + r, j*stride + s, co] += x[n,i,j,ci] * w[r,s,co,ci]
        out = full[:, before : before + stride*H, ...]
    For k=3, s=2: before=0 -> the LAST row/col of ``full`` is dropped.
    """
    xv = x.v
    N, H, W, Ci = xv.shape
    R, S, Co, Ciw = w.v.shape
    assert Ciw == Ci
    Ho, Wo = H * stride, W * stride
    pt, _, oh = same_pads(Ho, R, stride)
    pl, _, ow = same_pads(Wo, S, stride)
    assert oh == H and ow == W
    FH, FW = (H - 1) * stride + R, (W - 1) * stride + S
    full = np.zeros((N, FH, FW, Co))
    xm = xv.reshape(-1, Ci)
    for r in range(R):
        for s in range(S):
            full[:, r:r + (H - 1) * stride + 1:stride, s:s + (W - 1) * stride + 1:stride] += \
                (xm @ w.v[r, s].T).reshape(N, H, W, Co)
    y = Var(full[:, pt:pt + Ho, pl:pl + Wo] + b.v)

    def vjp(gy):
        gfull = np.zeros_like(full)
        gfull[:, pt:pt + Ho, pl:pl + Wo] = gy
        gw = np.zeros_like(w.v)
        gx = np.zeros((N * H * W, Ci))
        for r in range(R):
            for s in range(S):
                sl = gfull[:, r:r + (H - 1) * stride + 1:stride, s:s + (W - 1) * stride + 1:stride].reshape(-1, Co)
                gw[r, s] = sl.T @ xm
                gx += sl @ w.v[r, s]
        w.acc(gw)
        b.acc(gy.sum((0, 1, 2)))
        x.acc(gx.reshape(xv.shape))

    tape.record(y, vjp)
    return y


def instance_norm(tape, x: Var, gamma: Var, beta: Var, eps=1e-3) -> Var:
    """tfa.layers.InstanceNormalization() (module.py:212,216,233,...,308).

    [3P] tfa 0.9.1 GroupNormalization(groups=-1): per (n,c) mean and BIASED
    variance over H,W; y = gamma*(x-mean)/sqrt(var+eps)+beta, eps=1e-3,
    gamma=1, beta=0 at init.  ``ops.instance_norm`` (ops.py:13-22, dead code)
    is the same formula with eps=1e-5 -- pass eps to get it.
    """
    xv = x.v
    mu = xv.mean((1, 2), keepdims=True)
    var = ((xv - mu) ** 2).mean((1, 2), keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = (xv - mu) * rstd
    y = Var(gamma.v * xh + beta.v)

    def vjp(gy):
        gamma.acc((gy * xh).sum((0, 1, 2)))
        beta.acc(gy.sum((0, 1, 2)))
        gxh = gy * gamma.v
        gx = rstd * (gxh - gxh.mean((1, 2), keepdims=True) - xh * (gxh * xh).mean((1, 2), keepdims=True))
        x.acc(gx)

    tape.record(y, vjp)
    return y


class KinkPolicy:
    """Which side of a ReLU / LeakyReLU kink an element is on, when that is not decidable at float32 precision.

    The train step is piecewise linear in its activations: an element whose pre-activation is within float32 rounding of
    zero takes slope 1 in one correct implementation and slope 0 (0.3) in another, and that one difference reaches every
    gradient behind it at full size (tools/diag_d_f32.py: a pre-activation of 5e-7 moved D's gradients by 3e-4 of their
    norm).  With ~5 M activations per step some such elements always exist, so no seed avoids them.  Under a policy, elements
    with |pre-activation| < ``margin`` are AMBIGUOUS: for them the oracle takes the branch the implementation under test
    took (``branches[k]``: a boolean array per relu / lrelu call, in call order: True = the positive side); everywhere else it
    keeps its own float64 decision and counts a disagreement as an error of the implementation (``disagree_outside``).  The
    result is the exact float64 gradient of the same linear piece, which a float32 implementation must then match to
    float32 accuracy -- with the reference's own slopes."""

    def __init__(self, margin, branches=None):
        self.margin, self.branches = float(margin), branches
        self.calls = self.ambiguous = self.overridden = self.disagree_outside = self.elements = 0

    def decide(self, pre):
        pos = pre > 0
        amb = np.abs(pre) < self.margin
        k, self.calls = self.calls, self.calls + 1
        self.ambiguous += int(amb.sum())
        self.elements += pre.size
        if self.branches is not None:
            ext = np.asarray(self.branches[k], bool)
            assert ext.shape == pos.shape, (k, ext.shape, pos.shape)
            diff = ext != pos
            self.disagree_outside += int((diff & ~amb).sum())
            self.overridden += int((diff & amb).sum())
            pos = np.where(amb, ext, pos)
        return pos


KINKS = None        # a KinkPolicy while a test evaluates the oracle kink-aware; None: plain float64 decisions


def _positive(pre):
    return KINKS.decide(pre) if KINKS is not None else pre > 0


def relu(tape, x: Var) -> Var:
    """tf.keras.layers.Activation('relu') (module.py:213,234,238,242,256,260)."""
    pos = _positive(x.v)
    y = Var(np.where(pos, x.v, 0.0))
    tape.record(y, lambda gy: x.acc(gy * pos))
    return y


def lrelu(tape, x: Var, leak=0.3) -> Var:
    """tf.keras.layers.LeakyReLU() (module.py:285-309), [3P] default alpha=0.3.
    ``ops.lrelu`` (ops.py:36-37): max(x, leak*x) with leak=0.2 -- same function
    for 0<leak<1; pass leak to get it."""
    pos = _positive(x.v)
    y = Var(np.where(pos, x.v, leak * x.v))
    tape.record(y, lambda gy: x.acc(gy * np.where(pos, 1.0, leak)))
    return y


def tanh(tape, x: Var) -> Var:
    """tf.keras.layers.Activation('tanh') (module.py:265)."""
    y = Var(np.tanh(x.v))
    tape.record(y, lambda gy: x.acc(gy * (1.0 - y.v ** 2)))
    return y


def add(tape, a: Var, b: Var) -> Var:
    """``y + x`` residual (module.py:217)."""
    y = Var(a.v + b.v)

    def vjp(gy):
        a.acc(gy)
        b.acc(gy)

    tape.record(y, vjp)
    return y


def mask_reduce(tape, h4: Var, mask: np.ndarray) -> Var:
    """module.py:312-314: ``multiply([h4, mask])`` then ``reduce_sum(axis=-1,
    keepdims=True)``.  Broadcasts like Keras multiply: h4 (N,1,1,C) against a
    (N,hm,wm,C) mask gives (N,hm,wm,1) (the 128x128 case)."""
    prod = h4.v * mask
    y = Var(prod.sum(-1, keepdims=True))

    def vjp(gy):
        g = gy * mask
        # un-broadcast onto h4's shape
        for ax in (1, 2):
            if h4.v.shape[ax] == 1 and g.shape[ax] != 1:
                g = g.sum(ax, keepdims=True)
        h4.acc(g)

    tape.record(y, vjp)
    return y


def bce_logits_mean(tape, logits: Var, label: float) -> Var:
    """tf.keras.losses.BinaryCrossentropy(from_logits=True)(label*ones, logits)
    (model.py:150,153,161-163).  [3P] per-element max(x,0) - x*z + log1p(exp(-|x|)),
    reduced with a global mean (mean over last axis, then mean over the rest)."""
    x = logits.v
    per = np.maximum(x, 0) - x * label + np.log1p(np.exp(-np.abs(x)))
    y = Var(per.mean())
    sig = 1.0 / (1.0 + np.exp(-x))
    tape.record(y, lambda gy: logits.acc(gy * (sig - label) / x.size))
    return y


def l1_mean(tape, a: np.ndarray, b: Var) -> Var:
    """``tf.reduce_mean(tf.abs(seg_A - fake_A))`` (model.py:155); a is constant."""
    d = a - b.v
    y = Var(np.abs(d).mean())
    tape.record(y, lambda gy: b.acc(-gy * np.sign(d) / d.size))
    return y


def scale_add(tape, a: Var, b: Var, sb: float) -> Var:
    """a + sb*b on scalars (model.py:156 ``gan_loss + LAMBDA*l1_loss``; :164)."""
    y = Var(a.v + sb * b.v)

    def vjp(gy):
        a.acc(gy)
        b.acc(gy * sb)

    tape.record(y, vjp)
    return y


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------

def glorot_uniform(rng, shape):
    """[3P] Keras default kernel_initializer for Conv2D/Conv2DTranspose:
    U(-l, l), l = sqrt(6/(fan_in+fan_out)); fans = shape[-2]*rf, shape[-1]*rf."""
    rf = int(np.prod(shape[:-2]))
    lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
    return rng.uniform(-lim, lim, size=shape)


def generator_param_shapes(gf_dim=64, in_c=3, out_c=3, n_blocks=9):
    """Ordered (name, shape) list = Keras ``trainable_variables`` creation order of
    ``generator_resnet`` (module.py:219-269): kernel, bias, then IN gamma, beta."""
    L = []

    def conv(name, shape, norm=True):
        L.append((name + "_w", shape))
        L.append((name + "_b", (shape[-1] if not name.startswith("d") else shape[-2],)))
        if norm:
            c = L[-1][1][0]
            L.append((name + "_g", (c,)))
            L.append((name + "_beta", (c,)))

    conv("c1", (7, 7, in_c, gf_dim))
    conv("c2", (3, 3, gf_dim, gf_dim * 2))
    conv("c3", (3, 3, gf_dim * 2, gf_dim * 4))
    for i in range(1, n_blocks + 1):
        conv(f"r{i}a", (3, 3, gf_dim * 4, gf_dim * 4))
        conv(f"r{i}b", (3, 3, gf_dim * 4, gf_dim * 4))
    conv("d1", (3, 3, gf_dim * 2, gf_dim * 4))   # Conv2DTranspose: (kh,kw,out,in)
    conv("d2", (3, 3, gf_dim, gf_dim * 2))
    conv("out", (7, 7, gf_dim, out_c), norm=False)
    return L


def discriminator_param_shapes(df_dim=64, in_c=3, segment_class=34):
    """Creation order of ``discriminator`` (module.py:272-318)."""
    L = []

    def conv(name, shape, norm=True):
        L.append((name + "_w", shape))
        L.append((name + "_b", (shape[-1],)))
        if norm:
            L.append((name + "_g", (shape[-1],)))
            L.append((name + "_beta", (shape[-1],)))

    conv("h0", (3, 3, in_c, df_dim), norm=False)
    conv("h1", (3, 3, df_dim, df_dim * 2))
    conv("h2", (3, 3, df_dim * 2, df_dim * 4))
    conv("h3", (3, 3, df_dim * 4, df_dim * 8))
    conv("h31", (3, 3, df_dim * 8, df_dim * 8))
    conv("h32", (3, 3, df_dim * 8, df_dim * 8))
    conv("h33", (3, 3, df_dim * 8, df_dim * 8))
    conv("h4", (3, 3, df_dim * 8, segment_class), norm=False)
    return L


def init_params(shapes, rng, perturb=0.0):
    """[3P] Keras defaults: glorot_uniform kernels, zero bias, IN gamma=1 beta=0.
    perturb>0 adds N(0,perturb) to bias/gamma/beta so tests exercise them."""
    P = {}
    for name, shp in shapes:
        if name.endswith("_w"):
            P[name] = glorot_uniform(rng, shp)
        elif name.endswith("_g"):
            P[name] = np.ones(shp) + perturb * rng.standard_normal(shp)
        else:
            P[name] = perturb * rng.standard_normal(shp)
    return P


# ----------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------

def generator_resnet(tape, P, x: Var, n_blocks=9, eps=1e-3) -> Var:
    """module.py:219-269 (+ residule_block :208-217).  P: dict name -> Var."""

    def cin(name, h, stride=1, padding="VALID", reflect=0):
        h = conv2d(tape, h, P[name + "_w"], P[name + "_b"], stride, padding, reflect)
        return instance_norm(tape, h, P[name + "_g"], P[name + "_beta"], eps)

    h = relu(tape, cin("c1", x, reflect=3))                       # :230-234
    h = relu(tape, cin("c2", h, 2, "SAME"))                       # :236-238
    h = relu(tape, cin("c3", h, 2, "SAME"))                       # :240-242
    for i in range(1, n_blocks + 1):                              # :244-252
        y = relu(tape, cin(f"r{i}a", h, reflect=1))               # :210-213
        y = cin(f"r{i}b", y, reflect=1)                           # :214-216
        h = add(tape, y, h)                                       # :217
    for name in ("d1", "d2"):                                     # :254-260
        h = deconv2d(tape, h, P[name + "_w"], P[name + "_b"], 2)
        h = relu(tape, instance_norm(tape, h, P[name + "_g"], P[name + "_beta"], eps))
    h = conv2d(tape, h, P["out_w"], P["out_b"], 1, "VALID", 3)    # :262-264
    return tanh(tape, h)                                          # :265


def discriminator(tape, P, x: Var, mask: np.ndarray, leak=0.3, eps=1e-3) -> Var:
    """module.py:272-318."""
    h = lrelu(tape, conv2d(tape, x, P["h0_w"], P["h0_b"], 2, "SAME"), leak)   # :284-285
    for name, stride, pad in (("h1", 2, "SAME"), ("h2", 2, "SAME"), ("h3", 1, "SAME"),
                              ("h31", 2, "VALID"), ("h32", 2, "VALID"), ("h33", 1, "VALID")):
        h = conv2d(tape, h, P[name + "_w"], P[name + "_b"], stride, pad)       # :287-307
        h = lrelu(tape, instance_norm(tape, h, P[name + "_g"], P[name + "_beta"], eps), leak)
    h4 = conv2d(tape, h, P["h4_w"], P["h4_b"], 1, "SAME")                      # :311
    return mask_reduce(tape, h4, mask)                                         # :312-314


def disc_out_hw(H, W):
    """Spatial size of D's h4 map for an HxW input (module.py:284-311)."""
    def sz(n):
        for _ in range(3):
            n = -(-n // 2)          # h0,h1,h2 SAME s2
        n = (n - 3) // 2 + 1        # h31 VALID s2
        n = (n - 3) // 2 + 1        # h32 VALID s2
        n = n - 2                   # h33 VALID s1
        return n
    return sz(H), sz(W)


# ----------------------------------------------------------------------------
# optimizer + train step
# ----------------------------------------------------------------------------

def adam_tf(theta, g, m, v, t, lr=1e-3, beta1=0.5, beta2=0.999, eps=1e-7):
    """[3P] tf.keras.optimizers.Adam (model.py:205-207; beta1 from main.py:28):
        m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g^2
        lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  theta -= lr_t * m / (sqrt(v) + eps)
    (epsilon OUTSIDE the bias correction -- differs from torch.optim.Adam)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v


def train_step(PG, PD, real_A, seg_A, mask_A, opt_state=None, t=1, lr=1e-3, beta1=0.5,
               n_blocks=9, leak=0.3, eps=1e-3, l1_lambda=100.0):
    """model.py:169-200 with the documented deviations D2 (fake_A = G(real_A)
    every step; da_fake computed once -- :187 and :188 are the same values).

    PG, PD: dict name -> ndarray.  Returns dict with fake_A, gen_loss, disc_loss,
    grads (gG, gD) and the post-Adam parameters / optimizer slots.
    """
    tape = Tape()
    VG = {k: Var(v, k) for k, v in PG.items()}
    VD = {k: Var(v, k) for k, v in PD.items()}
    xA = Var(real_A)
    fake = generator_resnet(tape, VG, xA, n_blocks, eps)                 # :175-179
    seg = Var(seg_A)
    da_real = discriminator(tape, VD, seg, mask_A, leak, eps)            # :186
    da_fake = discriminator(tape, VD, fake, mask_A, leak, eps)           # :187 (=:188)
    gan = bce_logits_mean(tape, da_fake, 1.0)                            # :153
    l1 = l1_mean(tape, np.asarray(seg_A, F64), fake)                     # :155
    gen_loss = scale_add(tape, gan, l1, l1_lambda)                       # :156
    real_l = bce_logits_mean(tape, da_real, 1.0)                         # :162
    fake_l = bce_logits_mean(tape, da_fake, 0.0)                         # :163
    disc_loss = scale_add(tape, real_l, fake_l, 1.0)                     # :164

    def grads(loss, wrt):
        allv = [xA, seg] + list(VG.values()) + list(VD.values()) + [o for o, _ in tape.ops]
        for v in allv:
            v.g = None
        tape.backward([(loss, 1.0)])
        return {k: (np.zeros_like(v.v) if v.g is None else v.g.copy()) for k, v in wrt.items()}

    gG = grads(gen_loss, VG)      # :196 gen_tape.gradient(gen_loss, G vars)
    gD = grads(disc_loss, VD)     # :197 disc_tape.gradient(disc_loss, D vars)

    if opt_state is None:
        opt_state = {"mG": {k: np.zeros_like(v) for k, v in PG.items()},
                     "vG": {k: np.zeros_like(v) for k, v in PG.items()},
                     "mD": {k: np.zeros_like(v) for k, v in PD.items()},
                     "vD": {k: np.zeros_like(v) for k, v in PD.items()}}
    newG, newD = {}, {}
    st = {"mG": {}, "vG": {}, "mD": {}, "vD": {}}
    for k in PG:                                                          # :199
        newG[k], st["mG"][k], st["vG"][k] = adam_tf(PG[k], gG[k], opt_state["mG"][k], opt_state["vG"][k], t, lr, beta1)
    for k in PD:                                                          # :200
        newD[k], st["mD"][k], st["vD"][k] = adam_tf(PD[k], gD[k], opt_state["mD"][k], opt_state["vD"][k], t, lr, beta1)
    return {"fake_A": fake.v, "da_real": da_real.v, "da_fake": da_fake.v,
            "gen_loss": float(gen_loss.v), "disc_loss": float(disc_loss.v),
            "gG": gG, "gD": gD, "PG": newG, "PD": newD, "opt_state": st}


# ----------------------------------------------------------------------------
# defined-not-wired SG-GAN pieces and the cycle-mode step built from them (deviation D5)
# ----------------------------------------------------------------------------

SOBEL_X = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], F64)     # module.py:327-329
SOBEL_Y = np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], F64)     # module.py:330-332


def mse_const_mean(tape, x: Var, target: float) -> Var:
    """``mae_criterion(in_, target)`` (module.py:340-341): reduce_mean((in_-target)**2) -- the LSGAN criterion
    (squared error despite the name) against ones_like / zeros_like (model.py:121,127-128)."""
    d = x.v - target
    y = Var((d ** 2).mean())
    tape.record(y, lambda gy: x.acc(gy * 2 * d / d.size))
    return y


def seg_edge_weight(seg: np.ndarray) -> np.ndarray:
    """model.py:108-119: REFLECT-pad 1, depthwise central differences [[0,0,0],[-1,0,1],[0,0,0]] and its
    transpose (VALID), abs, reduce_sum over channels (keepdims), sign, abs -> (N,H,W,1) in {0,1}."""
    N, H, W, C = seg.shape
    ih, iw = reflect_index(H, 1), reflect_index(W, 1)
    sp = np.asarray(seg, F64)[:, ih][:, :, iw]
    dx = sp[:, 1:H + 1, 2:W + 2] - sp[:, 1:H + 1, 0:W]
    dy = sp[:, 2:H + 2, 1:W + 1] - sp[:, 0:H, 1:W + 1]
    s = (np.abs(dx) + np.abs(dy)).sum(-1, keepdims=True)
    return np.abs(np.sign(s))


def _sobel(x):
    """tf_deriv (module.py:325-334): depthwise correlation with gx, gy, padding SAME (zero); returns (gx, gy)."""
    N, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    gx = np.zeros_like(x); gy = np.zeros_like(x)
    for a in range(3):
        for b in range(3):
            sl = xp[:, a:a + H, b:b + W]
            gx += SOBEL_X[a, b] * sl; gy += SOBEL_Y[a, b] * sl
    return gx, gy


def _sobel_T(ggx, ggy):
    N, H, W, C = ggx.shape
    gp = np.zeros((N, H + 2, W + 2, C))
    for a in range(3):
        for b in range(3):
            gp[:, a:a + H, b:b + W] += SOBEL_X[a, b] * ggx + SOBEL_Y[a, b] * ggy
    return gp[:, 1:H + 1, 1:W + 1]


def gradloss(tape, in_: Var, target: np.ndarray, weight: np.ndarray) -> Var:
    """gradloss_criterion (module.py:347-351): abs_deriv = | |tf_deriv(in_)| - |tf_deriv(target)| |,
    reduce_mean over the 2C derivative channels (keepdims), reduce_mean(weight * abs_deriv)."""
    ax, ay = _sobel(in_.v)
    bx, by = _sobel(np.asarray(target, F64))
    dx, dy = np.abs(ax) - np.abs(bx), np.abs(ay) - np.abs(by)
    C = in_.v.shape[-1]
    per = (np.abs(dx) + np.abs(dy)).sum(-1, keepdims=True) / (2 * C)
    y = Var((weight * per).mean())

    def vjp(gy_):
        k = gy_ * weight / (2 * C) / weight.size
        in_.acc(_sobel_T(k * np.sign(dx) * np.sign(ax), k * np.sign(dy) * np.sign(ay)))

    tape.record(y, vjp)
    return y


def add_scalars(tape, terms) -> Var:
    """sum_i coef_i * term_i on scalar Vars."""
    y = Var(sum(c * t.v for c, t in terms))

    def vjp(gy):
        for c, t in terms:
            t.acc(gy * c)

    tape.record(y, vjp)
    return y


def cycle_step(PGab, PGba, PDa, PDb, real_A, real_B, seg_A, seg_B, mask_A, mask_B, opt_state=None, t=1,
               lr=2e-4, beta1=0.5, L1_lambda=10.0, Lg_lambda=5.0, use_lsgan=True, n_blocks=9, leak=0.3, eps=1e-3):
    """The 2G+2D SG-GAN step (north_star unit; deviation D5) assembled from the reference's defined-not-wired
    criteria: generator_loss / discriminator_loss (model.py:114-133) with criterionGAN = mae_criterion (use_lsgan,
    main.py:39) or sce_criterion, abs_criterion cycle terms weighted by --L1_lambda (main.py:37) and the
    gradient-sensitive terms weighted by --Lg_lambda (main.py:38) on the seg-edge indicator (model.py:108-119);
    Adam(lr=--lr, beta1) for all four networks.  Domain B fakes are judged on A's semantics (mask_A) and vice versa.
    No image pool (ImagePool is never called in the reference), so D sees the current fakes."""
    tape = Tape()
    V = lambda P: {k: Var(v, k) for k, v in P.items()}
    Gab, Gba, Da, Db = V(PGab), V(PGba), V(PDa), V(PDb)
    xA, xB = Var(real_A), Var(real_B)
    fake_B = generator_resnet(tape, Gab, xA, n_blocks, eps)
    cyc_A = generator_resnet(tape, Gba, fake_B, n_blocks, eps)
    fake_A = generator_resnet(tape, Gba, xB, n_blocks, eps)
    cyc_B = generator_resnet(tape, Gab, fake_A, n_blocks, eps)
    DB_fake = discriminator(tape, Db, fake_B, mask_A, leak, eps)
    DA_fake = discriminator(tape, Da, fake_A, mask_B, leak, eps)
    DA_real = discriminator(tape, Da, xA, mask_A, leak, eps)
    DB_real = discriminator(tape, Db, xB, mask_B, leak, eps)
    crit = (lambda x, z: mse_const_mean(tape, x, z)) if use_lsgan else (lambda x, z: bce_logits_mean(tape, x, z))
    wA, wB = seg_edge_weight(seg_A), seg_edge_weight(seg_B)
    g_loss = add_scalars(tape, [
        (1.0, crit(DA_fake, 1.0)), (1.0, crit(DB_fake, 1.0)),
        (L1_lambda, l1_mean(tape, np.asarray(real_A, F64), cyc_A)), (L1_lambda, l1_mean(tape, np.asarray(real_B, F64), cyc_B)),
        (Lg_lambda, gradloss(tape, fake_A, real_B, wB)), (Lg_lambda, gradloss(tape, fake_B, real_A, wA))])
    d_loss = add_scalars(tape, [(0.5, crit(DA_real, 1.0)), (0.5, crit(DA_fake, 0.0)),
                                (0.5, crit(DB_real, 1.0)), (0.5, crit(DB_fake, 0.0))])

    nets = {"Gab": Gab, "Gba": Gba, "Da": Da, "Db": Db}

    def grads(loss, which):
        for v in [xA, xB] + [p for n in nets.values() for p in n.values()] + [o for o, _ in tape.ops]:
            v.g = None
        tape.backward([(loss, 1.0)])
        return {n: {k: (np.zeros_like(v.v) if v.g is None else v.g.copy()) for k, v in nets[n].items()} for n in which}

    G = grads(g_loss, ("Gab", "Gba"))
    G.update(grads(d_loss, ("Da", "Db")))
    params = {"Gab": PGab, "Gba": PGba, "Da": PDa, "Db": PDb}
    if opt_state is None:
        opt_state = {n: {"m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}}
                     for n, P in params.items()}
    new, st = {}, {}
    for n, P in params.items():
        new[n], st[n] = {}, {"m": {}, "v": {}}
        for k in P:
            new[n][k], st[n]["m"][k], st[n]["v"][k] = adam_tf(P[k], G[n][k], opt_state[n]["m"][k], opt_state[n]["v"][k], t, lr, beta1)
    return {"fake_A": fake_A.v, "fake_B": fake_B.v, "cyc_A": cyc_A.v, "cyc_B": cyc_B.v, "g_loss": float(g_loss.v),
            "d_loss": float(d_loss.v), "grads": G, "params": new, "opt_state": st}


# ----------------------------------------------------------------------------
# next-row helpers: image pool, evaluation scores
# ----------------------------------------------------------------------------

class ImagePoolRef:
    """utils.py:27-53 restated on NumPy arrays; ``rng.rand()`` replaces the module-level ``np.random.rand()``."""

    def __init__(self, maxsize=50, rng=None):
        self.maxsize, self.num_img, self.images = maxsize, 0, []
        self.rng = rng if rng is not None else np.random

    def __call__(self, image):
        if self.maxsize <= 0:
            return image
        if self.num_img < self.maxsize:
            self.images.append([np.array(t) for t in image])
            self.num_img += 1
            return image
        if self.rng.rand() > 0.5:
            idx = int(self.rng.rand() * self.maxsize)
            tmp1, tmp3 = self.images[idx][0], self.images[idx][2]
            self.images[idx][0], self.images[idx][2] = np.array(image[0]), np.array(image[2])
            idx = int(self.rng.rand() * self.maxsize)
            tmp2, tmp4 = self.images[idx][1], self.images[idx][3]
            self.images[idx][1], self.images[idx][3] = np.array(image[1]), np.array(image[3])
            return [tmp1, tmp2, tmp3, tmp4]
        return image


def fast_hist(label_true, label_pred, n_class):
    """metric.py:18-24."""
    mask = (label_true >= 0) & (label_true < n_class)
    return np.bincount(n_class * label_true[mask].astype(int) + label_pred[mask], minlength=n_class ** 2).reshape(n_class, n_class)


def scores(label_trues, label_preds, n_class):
    """metric.py:27-47."""
    hist = np.zeros((n_class, n_class))
    for lt, lp in zip(label_trues, label_preds):
        hist += fast_hist(lt.flatten(), lp.flatten(), n_class)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        valid = hist.sum(axis=1) > 0
        mean_iu = np.nanmean(iu[valid])
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return {"Overall Acc": acc, "Mean Acc": acc_cls, "FreqW Acc": fwavacc, "Mean IoU": mean_iu, "Class IoU": dict(zip(range(n_class), iu))}


def scores_seg_fake(seg_image, fake_img):
    """metric.py:71-77 (fake_img already a NumPy array here)."""
    gts = np.argmax((255 * seg_image).astype(np.uint8).transpose(0, 3, 2, 1), axis=1)
    preds = np.argmax((255 * fake_img).astype(np.uint8).transpose(0, 3, 2, 1), axis=1)
    return gts, preds
