"""Network definitions of the SG-GAN hot path -- host-side mirror of the reference's ``module.py``.

``generator_resnet()`` (module.py:219-269, with ``residule_block`` :208-217) and
``discriminator()`` (module.py:272-318) keep the reference's names and call
conventions (``generator(x)``, ``discriminator([x, mask])``) but run on an explicit
forward/backward engine over the HIP kernels of libsggan.so: no tracing compiler, no
autograd graph in the training loop.  Each network owns ONE flat f32 buffer for its
parameters, one for gradients and two for the Adam slots, so the optimizer is a single
launch and data-parallel training all-reduces one bucket per network.

Deviation D3: ``gf_dim/df_dim/segment_class`` and the image size are
parameters (defaults = the module-local constants of the reference); layers are
shape-agnostic, geometry is resolved per call.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _abi as A
from . import kernels as K


# ----------------------------------------------------------------------------- parameters
class ParamStore:
    """Flat parameter storage.  1-D parameters (bias, gamma, beta) are stored padded to the
    channel granule (8) so kernels can index them by padded channel; padded entries stay 0."""

    def __init__(self, specs, device):
        self.specs = list(specs)
        self.index = {}
        off = 0
        for name, shape in self.specs:
            shape = tuple(int(s) for s in shape)
            n = K.cpad(shape[0]) if len(shape) == 1 else int(np.prod(shape))
            self.index[name] = (off, n, shape)
            off += (n + 3) // 4 * 4
        self.numel = off
        self.device = torch.device(device)
        z = lambda: torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.flat, self.grad, self.m, self.v = z(), z(), z(), z()
        self.version = 0            # bumped whenever parameter values change (re-pack trigger)
        # Adam's step number lives on the device (Keras: the `optimizer.iterations` variable), so the update launch
        # carries no host state and can be replayed from a captured HIP graph
        self._adam_state = torch.zeros(2, dtype=torch.int64, device=self.device)    # [iterations, scratch] (sgg_adam_iter)
        self.iterations = self._adam_state[0:1]

    @property
    def step_count(self):
        """Adam t (host copy of the device counter; synchronises -- checkpoint / test use only)."""
        return int(self.iterations.item())

    @step_count.setter
    def step_count(self, t):
        self.iterations.fill_(int(t))

    # views ------------------------------------------------------------------
    def _view(self, buf, name, padded):
        off, n, shape = self.index[name]
        if len(shape) == 1:
            return buf[off:off + n] if padded else buf[off:off + shape[0]]
        return buf[off:off + n].view(shape)

    def p(self, name, padded=True):
        return self._view(self.flat, name, padded)

    def g(self, name, padded=False, buf=None):
        return self._view(self.grad if buf is None else buf, name, padded)

    def names(self):
        return [n for n, _ in self.specs]

    def n_real(self):
        return sum(int(np.prod(s)) for _, s in self.specs)

    # init / io ----------------------------------------------------------------
    def init_keras(self, seed):
        """Keras defaults: glorot_uniform kernels, zero bias, IN gamma=1 / beta=0."""
        gen = torch.Generator(device="cpu").manual_seed(int(seed))
        host = torch.zeros(self.numel, dtype=torch.float32)
        for name, shape in self.specs:
            off, n, shape = self.index[name]
            if name.endswith("_w"):
                rf = int(np.prod(shape[:-2]))
                lim = math.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
                host[off:off + n] = (torch.rand(n, generator=gen) * 2 - 1) * lim
            elif name.endswith("_g"):
                host[off:off + shape[0]] = 1.0
        self.flat.copy_(host)
        self.version += 1

    def load(self, params: dict):
        """name -> array in the reference layouts (HWIO kernels, (kh,kw,out,in) transpose kernels)."""
        host = torch.zeros(self.numel, dtype=torch.float32)
        for name, _ in self.specs:
            off, n, shape = self.index[name]
            a = torch.as_tensor(np.asarray(params[name], dtype=np.float32)).reshape(-1)
            assert a.numel() == int(np.prod(shape)), (name, a.numel(), shape)
            host[off:off + a.numel()] = a
        self.flat.copy_(host)
        self.version += 1

    def export(self, buf=None) -> dict:
        src = (self.flat if buf is None else buf).detach().cpu()
        out = {}
        for name, _ in self.specs:
            off, n, shape = self.index[name]
            out[name] = src[off:off + int(np.prod(shape))].view(shape).numpy().copy()
        return out

    def zero_grad(self):
        self.grad.zero_()

    def adam_step(self, lr=1e-3, beta1=0.5, beta2=0.999, eps=1e-7, grad_scale=1.0):
        """tf.keras.optimizers.Adam.apply_gradients (model.py:199-200) over the whole network: one launch."""
        K.adam_iter(self.flat, self.grad, self.m, self.v, self._adam_state, lr, beta1, beta2, eps, grad_scale)
        self.version += 1


class Adam:
    """``tf.keras.optimizers.Adam`` as the reference holds it in ``self.g_optim`` / ``self.d_optim`` (model.py:83-84,
    199-200, 205-207), bound to one network's flat parameter store: the whole update is one fused launch."""

    def __init__(self, net, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.net, self.learning_rate, self.beta_1, self.beta_2, self.epsilon = net, learning_rate, beta_1, beta_2, epsilon

    @property
    def iterations(self):
        return self.net.P.iterations

    def apply_gradients(self, grads_and_vars=None, grad_scale=1.0):
        """``optimizer.apply_gradients(zip(grads, net.trainable_variables))`` (model.py:199-200).  Called without
        arguments it consumes the network's own gradient buffer (what ``sggan.train_step`` does); given (grad, var)
        pairs, the gradients are first copied over that buffer -- ``var`` must be one of ``net.trainable_variables``."""
        P = self.net.P
        if grads_and_vars is not None:
            names = P.names()
            by_ptr = {v.data_ptr(): n for n, v in zip(names, self.net.trainable_variables)}
            P.zero_grad()
            for g, v in grads_and_vars:
                name = by_ptr.get(v.data_ptr())
                if name is None:
                    raise ValueError("apply_gradients: variable does not belong to this optimizer's network")
                if g is not None:
                    P.g(name).copy_(torch.as_tensor(g, device=P.device).reshape(P.g(name).shape))
        P.adam_step(self.learning_rate, self.beta_1, self.beta_2, self.epsilon, grad_scale)


def generator_param_specs(gf_dim=64, in_c=3, out_c=3, n_blocks=9):
    """(name, shape) in Keras ``trainable_variables`` creation order of generator_resnet (module.py:219-269)."""
    L = []

    def conv(name, shape, out_ch, norm=True):
        L.append((name + "_w", shape))
        L.append((name + "_b", (out_ch,)))
        if norm:
            L.append((name + "_g", (out_ch,)))
            L.append((name + "_beta", (out_ch,)))

    conv("c1", (7, 7, in_c, gf_dim), gf_dim)
    conv("c2", (3, 3, gf_dim, gf_dim * 2), gf_dim * 2)
    conv("c3", (3, 3, gf_dim * 2, gf_dim * 4), gf_dim * 4)
    for i in range(1, n_blocks + 1):
        conv(f"r{i}a", (3, 3, gf_dim * 4, gf_dim * 4), gf_dim * 4)
        conv(f"r{i}b", (3, 3, gf_dim * 4, gf_dim * 4), gf_dim * 4)
    conv("d1", (3, 3, gf_dim * 2, gf_dim * 4), gf_dim * 2)     # Conv2DTranspose kernel: (kh,kw,out,in)
    conv("d2", (3, 3, gf_dim, gf_dim * 2), gf_dim)
    conv("out", (7, 7, gf_dim, out_c), out_c, norm=False)
    return L


def discriminator_param_specs(df_dim=64, in_c=3, segment_class=34):
    """Creation order of discriminator (module.py:272-318)."""
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


def plan_buckets(P, unit_names, n_buckets):
    """Contiguous ranges of a network's flat gradient buffer, cut at layer boundaries into ``n_buckets`` groups of about equal
    size (plus, when there is more than one, the small leading layers as a bucket of their own -- below):
    [(first_unit_name, lo, hi)] in flat (= forward) order, covering [0, P.numel) exactly.  Backward visits the layers in
    reverse, so a group's gradients are complete once the backward of its FIRST unit (weight gradient included) has been queued
    -- the point at which the data-parallel step launches that bucket's all-reduce."""
    starts = [P.index[n + "_w"][0] for n in unit_names]
    assert starts == sorted(starts) and starts[0] == 0, "units must be given in flat-buffer order"
    total = P.numel
    n_buckets = max(1, min(int(n_buckets), len(unit_names)))
    cuts = [0]
    for k in range(1, n_buckets):
        target = total * k / n_buckets
        i = min(range(len(unit_names)), key=lambda j: abs(starts[j] - target))
        if i > cuts[-1]:
            cuts.append(i)
    # the group that completes LAST (the first in flat order) has nothing of the backward pass left to hide behind: cut its
    # small leading layers (the generator's stem, <= 5 % of the buffer) off as a bucket of their own, so that the exposed
    # exchange is a latency-sized one and the bulk of the group travels under those layers' backward
    if len(cuts) > 1 or n_buckets > 1:
        first_end = cuts[1] if len(cuts) > 1 else len(unit_names)
        small = [j for j in range(1, first_end) if starts[j] <= 0.05 * total]
        if small:
            cuts.insert(1, small[-1])
    ends = cuts[1:] + [len(unit_names)]
    return [(unit_names[a], starts[a], total if b == len(unit_names) else starts[b]) for a, b in zip(cuts, ends)]


# ----------------------------------------------------------------------------- layer engine
# Defaults of two per-network switches (``sggan(fuse_in_stats=..., fuse_in_bwd=...)`` / ``net.fuse_in_stats`` / ``net.fuse_in_bwd``;
# nothing here reads the environment).  fuse_in_stats: the conv epilogue emits the following instance norm's per-chunk sums.
# fuse_in_bwd: its backward counterpart (data-gradient epilogue -> norm-backward sums) -- OFF: the epilogue has to read the norm
# input tile (and the skip gradient) at the end of a grid that has nothing to overlap the burst with.  Measured at the bench shape
# (tools/bench_conv.py --ops dgrad_add,dgrad_stats,in_bwd,in_bwd_partial): 8 images +6.6 us per data gradient against 12.9 us saved
# in the norm; 16 images (the paired cycle step's launch size) +23.4 against 22.7 -- nothing left.  Kept, parity-tested, as an
# opt-in of the one-network-at-a-time sequencing; it has no paired form.
FUSE_CONV_IN_STATS = True
FUSE_CONV_IN_BWD = False
# Round 4: the same statistics epilogue exists for the transposed layers (sgg_deconv2d_fwd_stats: the stride-2 halo kernel's own STATS build)
# and for the stem (sgg_conv2d_fwd_stats on the narrow-input kernel).  Both are parity-tested and OFF: in the cycle step the transposed
# layers' epilogue costs more than the 152 us of statistics passes it removes (-0.8 % on the step: that kernel has no registers to
# spare: 32 spilled VGPRs + a second pass over the accumulators), the stem's is neutral (profiles/r04_stats_epilogues.txt).
FUSE_DECONV_IN_STATS = False
FUSE_STEM_IN_STATS = False


class _ConvUnit:
    """One Conv2D / Conv2DTranspose call site, optionally followed by InstanceNorm (+act, +residual),
    or by a fused activation when there is no norm.  Stateless w.r.t. activations: forward returns a
    record that backward consumes, so one network can be applied several times per step."""

    def __init__(self, net, name, kind, stride=1, padding="VALID", reflect=0, norm=True, act=A.ACT_NONE, leak=0.0):
        self.net, self.name, self.kind = net, name, kind            # kind: "conv" | "deconv"
        self.stride, self.padding, self.reflect = stride, padding, reflect
        self.norm, self.act, self.leak = norm, act, leak
        shape = net.P.index[name + "_w"][2]
        self.R, self.S = shape[0], shape[1]
        if kind == "conv":
            self.cin, self.cout = shape[2], shape[3]
        else:                                                       # (kh,kw,out,in)
            self.cout, self.cin = shape[2], shape[3]
        self._packed = (-1, None, None)
        self._pending = None                                        # deferred weight-gradient operands (pair_wgrads)

    def flush_wgrad(self):
        """Run a deferred weight gradient whose partner never came."""
        if self._pending is not None:
            g, x, dxc = self._pending
            self._pending = None
            K.conv_wgrad(g, x, dxc, self.net.P.g(self.name + "_w"), accumulate=True)

    def geom(self, x):
        N, H, W, Cp = x.shape
        assert Cp == K.cpad(self.cin), (self.name, Cp, self.cin)
        if self.kind == "conv":
            return K.conv_geom(N, H, W, Cp, K.cpad(self.cout), self.R, self.S, self.stride, self.padding, self.reflect, x.dtype)
        return K.deconv_geom(N, H, W, Cp, K.cpad(self.cout), self.R, self.S, self.stride, x.dtype)

    def pack_dims(self):
        """(Cpad, Kpad) of the packed GEMM operands: for a deconv the equivalent conv has C = deconv out, K = deconv in
        (the stored (kh,kw,out,in) kernel is already that conv's HWIO)."""
        return (K.cpad(self.cin), K.cpad(self.cout)) if self.kind == "conv" else (K.cpad(self.cout), K.cpad(self.cin))

    def packed(self, dtype):
        P = self.net.P
        key = (P.version, dtype)
        if self._packed[0] != key:
            if self.net.conv_units():
                self.net.repack_all(dtype)               # every layer of the net in one launch
            else:
                Cp, Kp = self.pack_dims()
                wf, wd = K.pack_weights(P.p(self.name + "_w"), Cp, Kp, dtype)
                self._packed = (key, wf, wd)
        return self._packed[1], self._packed[2]

    def forward(self, x, residual=None, record_only=False, out=None):
        """out: the caller's buffer for the layer output (layers without a norm only: the generator's last layer writes the
        fakes straight into the discriminators' stacked input).
        record_only: the caller only wants the forward record (activation checkpointing re-runs a residual block's second
        conv for its backward pass: the block's output is not needed again) -- where the conv's epilogue delivers the norm's
        sums, the norm's apply pass is skipped (same statistics, bit for bit) and y is None."""
        P, n = self.net.P, self.name
        g = self.geom(x)
        wf, wd = self.packed(x.dtype)
        fused_act = A.ACT_NONE if self.norm else self.act
        if self.kind == "conv" and self.norm and g.stats_chunks and self.net.fuse_in_stats and (self.R != 7 or self.net.fuse_in_stats_stem):
            # the conv's epilogue emits the norm's per-chunk sums: the norm skips its own pass over the tensor
            xc, part = K.conv_fwd_stats(g, x, wf, P.p(n + "_b"))
            if record_only:
                return None, (g, x, xc, K.instnorm_finalize(part, xc.shape[1] * xc.shape[2], self.net.eps))
            y, stats = K.instnorm_fwd_partial(xc, part, P.p(n + "_g"), P.p(n + "_beta"), residual, self.net.eps, self.act, self.leak)
            return y, (g, x, xc, stats)
        assert out is None or not self.norm
        if self.kind == "deconv" and self.norm and g.stats_chunks and self.net.fuse_in_stats and self.net.fuse_in_stats_deconv:
            # Conv2DTranspose + InstanceNormalization (module.py:254-260): the stride-2 halo kernel's epilogue emits the norm's sums too
            xc, part = K.deconv_fwd_stats(g, x, wd, P.p(n + "_b"))
            y, stats = K.instnorm_fwd_partial(xc, part, P.p(n + "_g"), P.p(n + "_beta"), residual, self.net.eps, self.act, self.leak)
            return y, (g, x, xc, stats)
        if self.kind == "conv":
            xc = K.conv_fwd(g, x, wf, P.p(n + "_b"), fused_act, self.leak, out=out)
        else:
            xc = K.deconv_fwd(g, x, wd, P.p(n + "_b"), fused_act, self.leak, out=out)
        if not self.norm:
            return xc, (g, x, xc, None)
        y, stats = K.instnorm_fwd(xc, P.p(n + "_g"), P.p(n + "_beta"), residual, self.net.eps, self.act, self.leak)
        return y, (g, x, xc, stats)

    def weight_grad(self, g, x, dxc, gbuf=None):
        """dW += wgrad(x, dxc).  When the net is applied twice this step (pair_wgrads) the first application's operands are
        kept and both weight gradients run as ONE launch when the second arrives (one set of split slabs, one reduce)."""
        P, n = self.net.P, self.name
        if self.kind == "conv" and g.wgrad_pair and self.net.pair_wgrads and gbuf is None:
            if self._pending is None:
                self._pending = (g, x, dxc)
            else:
                g0, x0, d0 = self._pending
                self._pending = None
                if g0.x_shape == g.x_shape:
                    K.conv_wgrad_pair(g, x0, d0, x, dxc, P.g(n + "_w"), accumulate=True)
                else:
                    K.conv_wgrad(g0, x0, d0, P.g(n + "_w"), accumulate=True)
                    K.conv_wgrad(g, x, dxc, P.g(n + "_w"), accumulate=True)
        else:
            (K.conv_wgrad if self.kind == "conv" else K.deconv_wgrad)(g, x, dxc, P.g(n + "_w", buf=gbuf), accumulate=True)

    def backward(self, rec, dy, want_dx=True, param_grads=True, gbuf=None, addend=None, dy_partial=None, next_norm=None):
        """dy_partial: the norm-backward partial sums of THIS unit's norm, already computed by the data gradient that
        produced dy.  next_norm = (unit, rec) of the layer that will consume the returned dx: if its norm's first pass can
        be done in this unit's data-gradient epilogue the call returns (dx, partial) instead of dx."""
        P, n = self.net.P, self.name
        g, x, xc, stats = rec
        wf, wd = self.packed(x.dtype)
        if self.norm:
            if param_grads:
                dg, db = P.g(n + "_g", buf=gbuf), P.g(n + "_beta", buf=gbuf)
            else:   # gradients w.r.t. gamma/beta not wanted: send them to scratch
                dg = db = self.net.scratch_vec(K.cpad(self.cout))
            if dy_partial is not None:
                dxc = K.instnorm_bwd_partial(dy, xc, dy_partial, P.p(n + "_g"), P.p(n + "_beta"), stats, dg, db, param_grads, self.act, self.leak)
            else:
                dxc = K.instnorm_bwd(dy, xc, P.p(n + "_g"), P.p(n + "_beta"), stats, dg, db, param_grads, self.act, self.leak)
            # the conv bias feeds an InstanceNorm: its gradient is identically 0 -> left at 0
        else:
            if dy.dtype != xc.dtype:
                dy = dy.to(xc.dtype)
            dxc = K.act_bwd(dy, xc, self.act, self.leak) if self.act != A.ACT_NONE else dy
            if param_grads:
                K.bias_grad(dxc, P.g(n + "_b", buf=gbuf), accumulate=True)
        if param_grads:
            self.weight_grad(g, x, dxc, gbuf)
        if not want_dx:
            return None
        if self.kind == "conv":
            # mixed mode: the gradient handed to the NEXT norm backward stays f32 (bf16 operands in the GEMM, f32 result)
            out_f32 = bool(self.net.mixed and next_norm is not None and next_norm[0].norm and g.dgrad_mixed and dxc.dtype == torch.bfloat16)
            if addend is not None and addend.dtype != dxc.dtype and not out_f32:
                addend = addend.to(dxc.dtype)
            if next_norm is not None:
                nu, nrec = next_norm
                if (not out_f32 and self.net.fuse_in_stats and self.net.fuse_in_bwd and g.bwd_stats_chunks and nu.norm
                        and tuple(nrec[2].shape) == g.x_shape):
                    NP = nu.net.P
                    return K.conv_dgrad_stats(g, dxc, wd, addend, nrec[2], nrec[3], NP.p(nu.name + "_g"), NP.p(nu.name + "_beta"),
                                              nu.act, nu.leak)
                return K.conv_dgrad(g, dxc, wd, addend, out_f32=out_f32), None
            return K.conv_dgrad(g, dxc, wd, addend)
        dx = K.deconv_dgrad(g, dxc, wf)
        dx = dx if addend is None else K.add(dx, addend.to(dx.dtype))
        return (dx, None) if next_norm is not None else dx


class _Net:
    def __init__(self, specs, dtype, device, eps, seed):
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("sggan networks run on the MI355X HIP path only (device must be a cuda/hip device)")
        A.lib()                                      # fail loudly if the extension is missing
        self.eps = eps
        self.P = ParamStore(specs, self.device)
        if seed is not None:
            self.P.init_keras(seed)
        self._scratch = None
        self._tv = None
        # set by a step that applies this net exactly twice (the cycle step): weight gradients of layers that support
        # it are deferred at the first backward and run with the second as one launch (_ConvUnit.backward)
        self.pair_wgrads = False
        # mixed precision (bf16 networks only): data gradients that feed an instance-norm backward are kept in f32 where the
        # kernel supports it (the residual chain) -- see sgg_conv2d_bwd_data_mixed in include/sggan.h
        self.mixed = False
        self.fuse_in_stats, self.fuse_in_bwd = FUSE_CONV_IN_STATS, FUSE_CONV_IN_BWD
        self.fuse_in_stats_deconv, self.fuse_in_stats_stem = FUSE_DECONV_IN_STATS, FUSE_STEM_IN_STATS
        self.group2 = True           # lockstep pairs: both networks' generic convolutions in one grouped launch (sgg_*_group2)
        # activation checkpointing (BASELINE.json configs[4]): a generator keeps only each residual block's INPUT and re-runs
        # the block's two convs + norms in backward (Generator.forward / _block_records)
        self.checkpoint_blocks = False
        self._pack_tables = {}

    def conv_units(self):
        return ()

    def repack_all(self, dtype):
        """Pack the GEMM operands of every conv layer from the current parameters with ONE launch (all of them go stale
        together at each optimizer step); buffers and the device-side item table are built once per dtype."""
        units = self.conv_units()
        st = self._pack_tables.get(dtype)
        if st is None:
            entries = [(self.P.p(u.name + "_w"),) + u.pack_dims() + (True, True) for u in units]
            st = K.pack_weights_batch(entries, dtype, self.device)
            self._pack_tables[dtype] = st
        table, n, max_elems, bufs = st
        K.repack_batch(table, n, max_elems, dtype)
        key = (self.P.version, dtype)
        for u, (wf, wd) in zip(units, bufs):
            u._packed = (key, wf, wd)

    def flush_wgrads(self):
        for u in self.conv_units():
            u.flush_wgrad()

    def pending_wgrads(self):
        return [u.name for u in self.conv_units() if u._pending is not None]

    def bucket_plan(self, n_buckets):
        """plan_buckets() over this network's layers."""
        return plan_buckets(self.P, [u.name for u in self.conv_units()], n_buckets)

    def scratch_vec(self, n):
        if self._scratch is None or self._scratch.numel() < n:
            self._scratch = torch.zeros(max(n, 1024), dtype=torch.float32, device=self.device)
        return self._scratch[:n]

    # reference-style accessors (model.py:196-200 use .trainable_variables)
    @property
    def trainable_variables(self):
        """Real-shaped views into the flat parameter buffer, in Keras creation order."""
        if self._tv is None:
            self._tv = [self.P.p(n, padded=False) for n in self.P.names()]
        return self._tv

    def requires_grad_(self, flag=True):
        """Opt the parameter views into torch autograd (only needed by tape-style callers of __call__)."""
        for v in self.trainable_variables:
            v.requires_grad_(flag)
        return self

    def to_internal(self, x, out=None):
        """(N,H,W,C_real) float32 -> channel-padded activation tensor in the network dtype (no-op if already internal).
        out: write (or copy) into the caller's buffer -- a slice of a stacked batch."""
        if x.dtype == self.dtype and x.shape[-1] % A.CPAD == 0:
            return x if out is None else out.copy_(x)
        return K.pad_channels(x.to(device=self.device, dtype=torch.float32).contiguous(), K.cpad(x.shape[-1]), self.dtype, out=out)


class _BlockInput:
    """Tape entry of a residual block under activation checkpointing: only the block's input tensor."""
    __slots__ = ("x",)

    def __init__(self, x):
        self.x = x


class Generator(_Net):
    """generator_resnet (module.py:219-269): c7s1-64, d128, d256, 9 x R256, u128, u64, c7s1-3 + tanh."""

    def __init__(self, gf_dim=64, in_c=3, out_c=3, n_blocks=9, dtype=torch.bfloat16, device="cuda", eps=1e-3, seed=19):
        super().__init__(generator_param_specs(gf_dim, in_c, out_c, n_blocks), dtype, device, eps, seed)
        self.in_c, self.out_c, self.n_blocks = in_c, out_c, n_blocks
        U = lambda *a, **k: _ConvUnit(self, *a, **k)
        self.c1 = U("c1", "conv", reflect=3, act=A.ACT_RELU)                       # :230-234
        self.c2 = U("c2", "conv", stride=2, padding="SAME", act=A.ACT_RELU)        # :236-238
        self.c3 = U("c3", "conv", stride=2, padding="SAME", act=A.ACT_RELU)        # :240-242
        self.blocks = [(U(f"r{i}a", "conv", reflect=1, act=A.ACT_RELU),            # residule_block :208-217
                        U(f"r{i}b", "conv", reflect=1, act=A.ACT_NONE)) for i in range(1, n_blocks + 1)]
        self.d1 = U("d1", "deconv", stride=2, act=A.ACT_RELU)                      # :254-256
        self.d2 = U("d2", "deconv", stride=2, act=A.ACT_RELU)                      # :258-260
        self.out = U("out", "conv", reflect=3, norm=False, act=A.ACT_TANH)         # :262-265

    def conv_units(self):
        return [self.c1, self.c2, self.c3] + [u for pair in self.blocks for u in pair] + [self.d1, self.d2, self.out]

    def forward(self, x, out=None):
        """x: internal (N,H,W,8).  Returns (fake internal (N,H,W,8), tape).  out: buffer for the result."""
        tape = []
        h = x
        for u in (self.c1, self.c2, self.c3):
            h, r = u.forward(h)
            tape.append(r)
        for ua, ub in self.blocks:
            y, ra = ua.forward(h)
            hin = h
            h, rb = ub.forward(y, residual=h)                 # IN(conv(y)) + x   (:216-217)
            tape.append(_BlockInput(hin) if self.checkpoint_blocks else (ra, rb))
        for u in (self.d1, self.d2, self.out):
            h, r = u.forward(h, out=out if u is self.out else None)
            tape.append(r)
        return h, tape

    def _block_records(self, k, rec):
        """(ra, rb) of residual block k: as saved by forward, or -- activation checkpointing -- recomputed from the block's
        saved input (the kernels are bitwise reproducible, so the records, and with them every gradient, are the same bits)."""
        if not isinstance(rec, _BlockInput):
            return rec
        ua, ub = self.blocks[k]
        y, ra = ua.forward(rec.x)
        _, rb = ub.forward(y, residual=rec.x, record_only=True)
        return ra, rb

    def backward(self, tape, dy, want_dx=False, param_grads=True, gbuf=None, on_unit_done=None, addend=None):
        """on_unit_done(name): called after each layer's backward (its weight gradient included) has been queued -- the
        data-parallel step hangs its per-bucket all-reduce launches on it (``bucket_plan``).  addend: a tensor shaped like the
        input gradient, added to it in the first layer's data-gradient store (the step's gradient joins: no extra pass)."""
        nb = self.n_blocks
        done = on_unit_done if on_unit_done is not None else (lambda name: None)
        d = dy
        for u, r in zip((self.out, self.d2, self.d1), (tape[5 + nb], tape[4 + nb], tape[3 + nb])):
            d = u.backward(r, d, True, param_grads, gbuf)
            done(u.name)
        # residual blocks: each data gradient also makes the first pass of the norm backward that consumes it (the
        # norm of the conv before it in forward order), so that norm skips its statistics pass over the tensor
        part = None
        blocks = list(reversed(self.blocks))
        cur = self._block_records(nb - 1, tape[2 + nb]) if nb else None
        for k, (ua, ub) in enumerate(blocks):
            ra, rb = cur
            # (the records of the block in front: its second norm consumes this block's data gradient -- under checkpointing
            # they are recomputed one block ahead, so at most two blocks' activations are alive)
            cur = self._block_records(nb - 2 - k, tape[1 + nb - k]) if k + 1 < nb else None
            t, pa = ub.backward(rb, d, True, param_grads, gbuf, dy_partial=part, next_norm=(ua, ra))
            done(ub.name)
            nxt = (blocks[k + 1][1], cur[1]) if k + 1 < nb else (self.c3, tape[2])
            d, part = ua.backward(ra, t, True, param_grads, gbuf, addend=d, dy_partial=pa, next_norm=nxt)   # + skip gradient (fused)
            done(ua.name)
            ra = rb = None
        d = self.c3.backward(tape[2], d, True, param_grads, gbuf, dy_partial=part)
        done("c3")
        d = self.c2.backward(tape[1], d, True, param_grads, gbuf)
        done("c2")
        d = self.c1.backward(tape[0], d, want_dx, param_grads, gbuf, addend=addend)
        done("c1")
        return d

    def __call__(self, x):
        """Drop-in for ``self.generator(self.real_A)`` (model.py:175): NHWC float32 in, NHWC float32 out."""
        return _apply_net(self, x)

    def _run(self, x):
        y, tape = self.forward(self.to_internal(x))
        return K.unpad_channels(y, self.out_c), tape

    def _run_backward(self, tape, dy_real, gbuf, want_dx):
        dy = K.pad_channels(dy_real.contiguous(), K.cpad(self.out_c), self.dtype)
        dx = self.backward(tape, dy, want_dx, True, gbuf)
        return None if dx is None else K.unpad_channels(dx, self.in_c)


class Discriminator(_Net):
    """discriminator (module.py:272-318): 8 convs, mask multiply, channel sum."""

    def __init__(self, df_dim=64, in_c=3, segment_class=34, dtype=torch.bfloat16, device="cuda", eps=1e-3, leak=0.3, seed=20):
        super().__init__(discriminator_param_specs(df_dim, in_c, segment_class), dtype, device, eps, seed)
        self.in_c, self.segment_class = in_c, segment_class
        U = lambda *a, **k: _ConvUnit(self, *a, act=A.ACT_LRELU, leak=leak, **k)
        self.units = [U("h0", "conv", stride=2, padding="SAME", norm=False),      # :284-285
                      U("h1", "conv", stride=2, padding="SAME"),                  # :287-289
                      U("h2", "conv", stride=2, padding="SAME"),                  # :291-293
                      U("h3", "conv", stride=1, padding="SAME"),                  # :295-297
                      U("h31", "conv", stride=2, padding="VALID"),                # :299-301
                      U("h32", "conv", stride=2, padding="VALID"),                # :303-305
                      U("h33", "conv", stride=1, padding="VALID")]                # :307-309
        self.h4 = _ConvUnit(self, "h4", "conv", stride=1, padding="SAME", norm=False, act=A.ACT_NONE)   # :311

    def conv_units(self):
        return list(self.units) + [self.h4]

    def forward(self, x, mask):
        """x internal (N,H,W,8); mask f32 (N,mh,mw,segment_class).  Returns (logits f32 (N,mh,mw,1), tape)."""
        tape = []
        h = x
        for u in self.units:
            h, r = u.forward(h)
            tape.append(r)
        h4, r = self.h4.forward(h)
        tape.append(r)
        mask = mask.to(device=self.device, dtype=torch.float32).contiguous()
        out = K.mask_reduce_fwd(h4, mask, self.segment_class)                    # :312-314
        tape.append((mask, tuple(h4.shape)))
        return out, tape

    def backward(self, tape, dlogits, want_dx=False, param_grads=True, gbuf=None, addend=None):
        """addend: added to the image gradient in h0's data-gradient store (Generator.backward)."""
        mask, h4_shape = tape[-1]
        d = K.mask_reduce_bwd(dlogits.contiguous(), mask, h4_shape, self.dtype, self.segment_class)
        d = self.h4.backward(tape[-2], d, True, param_grads, gbuf)
        for i in range(len(self.units) - 1, -1, -1):
            d = self.units[i].backward(tape[i], d, want_dx or i > 0, param_grads, gbuf, addend=addend if i == 0 else None)
        return d

    def slice_tape(self, tape, lo, hi):
        """The forward records of images [lo, hi) of a pass, as views (a backward pass through part of the batch: the step sends
        reals and fakes through D as one stacked pass, the generator's loss only needs the fakes)."""
        out = []
        for u, (g, x, xc, stats) in zip(self.conv_units(), tape[:-1]):
            xs = x[lo:hi]
            out.append((u.geom(xs), xs, xc[lo:hi], None if stats is None else stats[lo:hi]))
        mask, h4_shape = tape[-1]
        out.append((mask[lo:hi], (hi - lo,) + tuple(h4_shape[1:])))
        return out

    def out_hw(self, H, W):
        """Spatial size of the h4 map for an HxW input (deviation D1: the mask grid to use above 128x128)."""
        def sz(n):
            for _ in range(3):
                n = -(-n // 2)
            n = (n - 3) // 2 + 1
            n = (n - 3) // 2 + 1
            return n - 2
        return sz(H), sz(W)

    def __call__(self, inputs):
        """Drop-in for ``self.discriminator([image, mask])`` (model.py:186-188)."""
        x, mask = inputs
        return _apply_net(self, x, mask)

    def _run(self, x, mask):
        return self.forward(self.to_internal(x), mask)

    def _run_backward(self, tape, dlogits, gbuf, want_dx):
        dx = self.backward(tape, dlogits, want_dx, True, gbuf)
        return None if dx is None else K.unpad_channels(dx, self.in_c)


# ----------------------------------------------------------------------------- two networks in lockstep
class _PairUnit:
    """The same call site of TWO networks of one shape (G_A->B / G_B->A, D_A / D_B), applied to a batch that stacks their
    activations: images [:n] belong to ``ua``'s network, [n:] to ``ub``'s.  The convolutions run per half with each network's
    weights (into slices of one stacked output); the instance norm -- per image -- runs ONCE over the stacked tensor with the
    affine parameters picked by image index (sgg_instnorm_*_pair): twice the bytes per launch (a 67 MB pass reaches 5-5.5 TB/s,
    a 33 MB one 4.1-4.5), half the launches, and per image exactly the arithmetic of two separate calls."""

    def __init__(self, ua, ub):
        assert (ua.kind, ua.stride, ua.padding, ua.reflect, ua.norm, ua.act, ua.leak, ua.R, ua.cin, ua.cout) == \
               (ub.kind, ub.stride, ub.padding, ub.reflect, ub.norm, ub.act, ub.leak, ub.R, ub.cin, ub.cout)
        self.ua, self.ub = ua, ub

    def forward(self, x, residual=None, record_only=False, out=None):
        ua, ub = self.ua, self.ub
        assert out is None or not ua.norm
        n = x.shape[0] // 2
        halves = ((ua, slice(0, n)), (ub, slice(n, 2 * n)))
        g = ua.geom(x[:n])
        na, nb = ua.name, ub.name
        PA, PB = ua.net.P, ub.net.P
        fused_act = A.ACT_NONE if ua.norm else ua.act
        g2 = ua.geom(x) if ua.kind == "conv" else None     # geometry of the stacked batch
        if ua.kind == "conv" and ua.norm and g.stats_chunks and ua.net.fuse_in_stats and g2.pair_ok:
            # one launch for both networks (per-image weights): 512 blocks, the second round's halo loads run under the
            # first round's stores
            wfa, _ = ua.packed(x.dtype)
            wfb, _ = ub.packed(x.dtype)
            xc, part = K.conv_fwd_stats_pair(g2, x, wfa, PA.p(na + "_b"), wfb, PB.p(nb + "_b"), n)
            if record_only:          # (_ConvUnit.forward: the statistics without the apply pass)
                return None, (g, x, xc, K.instnorm_finalize(part, xc.shape[1] * xc.shape[2], ua.net.eps))
            y, stats = K.instnorm_fwd_partial_pair(xc, part, PA.p(na + "_g"), PA.p(na + "_beta"), PB.p(nb + "_g"), PB.p(nb + "_beta"), n,
                                                   residual, ua.net.eps, ua.act, ua.leak)
            return y, (g, x, xc, stats)
        if ua.kind == "conv" and ua.norm and g.stats_chunks and ua.net.fuse_in_stats and (ua.R != 7 or ua.net.fuse_in_stats_stem):
            xc = torch.empty((2 * n,) + tuple(g.y_shape[1:]), dtype=x.dtype, device=x.device)
            part = torch.empty((2 * n, g.stats_chunks, g.y_shape[3], 2), dtype=torch.float32, device=x.device)
            for u, sl in halves:
                wf, _ = u.packed(x.dtype)
                K.conv_fwd_stats(g, x[sl], wf, u.net.P.p(u.name + "_b"), out=xc[sl], out_partial=part[sl])
            y, stats = K.instnorm_fwd_partial_pair(xc, part, PA.p(na + "_g"), PA.p(na + "_beta"), PB.p(nb + "_g"), PB.p(nb + "_beta"), n,
                                                   residual, ua.net.eps, ua.act, ua.leak)
            return y, (g, x, xc, stats)
        if ua.kind == "deconv" and ua.norm and ua.net.fuse_in_stats and ua.net.fuse_in_stats_deconv and ua.net.group2:
            gs = ua.geom(x)                                    # the stacked batch: per-image weights in the stride-2 halo kernel
            if gs.stats_chunks:
                _, wda = ua.packed(x.dtype)
                _, wdb = ub.packed(x.dtype)
                xc, part = K.deconv_fwd_stats(gs, x, wda, PA.p(na + "_b"), pair=(wdb, PB.p(nb + "_b"), n))
                y, stats = K.instnorm_fwd_partial_pair(xc, part, PA.p(na + "_g"), PA.p(na + "_beta"), PB.p(nb + "_g"), PB.p(nb + "_beta"), n,
                                                       residual, ua.net.eps, ua.act, ua.leak)
                return y, (g, x, xc, stats)
        # every other convolution: ONE grouped launch for both networks (sgg_*_group2: each network's call as its own group of
        # blocks -- bit-identical to two calls; the discriminators' small maps are latency bound, two half-size launches cost twice)
        wfa, wda = ua.packed(x.dtype)
        wfb, wdb = ub.packed(x.dtype)
        if not ua.net.group2:                                 # A/B switch: one launch per network into slices of the stacked output
            xc = K._out(out, (2 * n,) + tuple(g.y_shape[1:]), x.dtype, x.device)
            for u, sl, wf, wd in ((ua, halves[0][1], wfa, wda), (ub, halves[1][1], wfb, wdb)):
                if u.kind == "conv":
                    K.conv_fwd(g, x[sl], wf, u.net.P.p(u.name + "_b"), fused_act, u.leak, out=xc[sl])
                else:
                    K.deconv_fwd(g, x[sl], wd, u.net.P.p(u.name + "_b"), fused_act, u.leak, out=xc[sl])
        elif ua.kind == "conv":
            xc = K.conv_fwd_group2(g, x, wfa, PA.p(na + "_b"), wfb, PB.p(nb + "_b"), fused_act, ua.leak, out=out)
        else:
            assert out is None
            xc = K.deconv_fwd_group2(g, x, wda, PA.p(na + "_b"), wdb, PB.p(nb + "_b"), fused_act, ua.leak)
        if not ua.norm:
            return xc, (g, x, xc, None)
        y, stats = K.instnorm_fwd_pair(xc, PA.p(na + "_g"), PA.p(na + "_beta"), PB.p(nb + "_g"), PB.p(nb + "_beta"), n,
                                       residual, ua.net.eps, ua.act, ua.leak)
        return y, (g, x, xc, stats)

    def backward(self, rec, dy, want_dx=True, param_grads=True, addend=None):
        ua, ub = self.ua, self.ub
        g, x, xc, stats = rec
        n = x.shape[0] // 2
        halves = ((ua, slice(0, n)), (ub, slice(n, 2 * n)))
        na, nb = ua.name, ub.name
        PA, PB = ua.net.P, ub.net.P
        if ua.norm:
            if param_grads:
                grads = (PA.g(na + "_g"), PA.g(na + "_beta"), PB.g(nb + "_g"), PB.g(nb + "_beta"))
            else:                                         # gradients w.r.t. gamma / beta not wanted: send them to scratch
                sa, sb = ua.net.scratch_vec(K.cpad(ua.cout)), ub.net.scratch_vec(K.cpad(ub.cout))
                grads = (sa, sa, sb, sb)
            dxc = K.instnorm_bwd_pair(dy, xc, PA.p(na + "_g"), PA.p(na + "_beta"), PB.p(nb + "_g"), PB.p(nb + "_beta"), n, stats,
                                      *grads, param_grads, ua.act, ua.leak)
            # (the conv bias in front of an instance norm has an identically zero gradient: left at 0)
        else:
            dxc = K.act_bwd(dy, xc, ua.act, ua.leak) if ua.act != A.ACT_NONE else dy
            if param_grads and ua.net.group2:
                K.bias_grad_group2(dxc, PA.g(na + "_b"), PB.g(nb + "_b"), accumulate=True)
            elif param_grads:
                for u, sl in halves:
                    K.bias_grad(dxc[sl], u.net.P.g(u.name + "_b"), accumulate=True)
        if param_grads:
            # both networks have this layer's first application waiting (pair_wgrads): all four weight gradients in ONE launch
            pa, pb = ua._pending, ub._pending
            quad = (ua.kind == "conv" and g.wgrad_pair and ua.net.pair_wgrads and ub.net.pair_wgrads and pa is not None and pb is not None
                    and pa[0].x_shape == g.x_shape and pb[0].x_shape == g.x_shape)
            defer = ua.kind == "conv" and g.wgrad_pair and (ua.net.pair_wgrads or ub.net.pair_wgrads)   # the 3x3 layers' two-application form
            if quad and K.conv_wgrad_pair2(g, (pa[1], pa[2], x[halves[0][1]], dxc[halves[0][1]], PA.g(na + "_w")),
                                           (pb[1], pb[2], x[halves[1][1]], dxc[halves[1][1]], PB.g(nb + "_w")), accumulate=True):
                ua._pending = ub._pending = None
            elif ua.net.group2 and not defer and pa is None and pb is None:
                # both networks' weight gradients as one grouped call: two main kernels, one slab reduce (bit-identical to two calls)
                K.conv_wgrad_group2(g, x, dxc, PA.g(na + "_w"), PB.g(nb + "_w"), accumulate=True)
            else:
                for u, sl in halves:
                    u.weight_grad(g, x[sl], dxc[sl])
        if not want_dx:
            return None
        if ua.kind == "conv" and ua.geom(x).pair_ok:
            return K.conv_dgrad_pair(ua.geom(x), dxc, ua.packed(x.dtype)[1], ub.packed(x.dtype)[1], n, addend)
        wfa, wda = ua.packed(x.dtype)
        wfb, wdb = ub.packed(x.dtype)
        if not ua.net.group2:
            dx = torch.empty((2 * n,) + tuple(g.x_shape[1:]), dtype=dxc.dtype, device=dxc.device)
            for u, sl, wf, wd in ((ua, halves[0][1], wfa, wda), (ub, halves[1][1], wfb, wdb)):
                if u.kind == "conv":
                    K.conv_dgrad(g, dxc[sl], wd, None if addend is None else addend[sl], out=dx[sl])
                else:
                    K.deconv_dgrad(g, dxc[sl], wf, out=dx[sl])
            return dx if (ua.kind == "conv" or addend is None) else K.add(dx, addend)
        if ua.kind == "conv":
            return K.conv_dgrad_group2(g, dxc, wda, wdb, addend)
        dx = K.deconv_dgrad_group2(g, dxc, wfa, wfb)
        return dx if addend is None else K.add(dx, addend)


class GeneratorPair:
    """generator_resnet (module.py:219-269) of two generators in lockstep: forward([x_a; x_b]) = [G_a(x_a); G_b(x_b)]."""

    def __init__(self, ga, gb):
        assert ga.n_blocks == gb.n_blocks
        self.a, self.b = ga, gb
        P = _PairUnit
        self.head = [P(ga.c1, gb.c1), P(ga.c2, gb.c2), P(ga.c3, gb.c3)]
        self.blocks = [(P(xa, xb), P(ya, yb)) for (xa, ya), (xb, yb) in zip(ga.blocks, gb.blocks)]
        self.tail = [P(ga.d1, gb.d1), P(ga.d2, gb.d2), P(ga.out, gb.out)]

    def forward(self, x, out=None):
        """out: buffer for the stacked result (the step hands in the middle of the discriminators' stacked input)."""
        tape, h = [], x
        for u in self.head:
            h, r = u.forward(h)
            tape.append(r)
        ckpt = self.a.checkpoint_blocks or self.b.checkpoint_blocks
        for ua, ub in self.blocks:
            y, ra = ua.forward(h)
            hin = h
            h, rb = ub.forward(y, residual=h)             # IN(conv(y)) + x   (module.py:216-217)
            tape.append(_BlockInput(hin) if ckpt else (ra, rb))
        for u in self.tail:
            h, r = u.forward(h, out=out if u is self.tail[-1] else None)
            tape.append(r)
        return h, tape

    def _block_records(self, k, rec):
        """Generator._block_records for the pair: recompute a checkpointed block from its saved (stacked) input."""
        if not isinstance(rec, _BlockInput):
            return rec
        ua, ub = self.blocks[k]
        y, ra = ua.forward(rec.x)
        _, rb = ub.forward(y, residual=rec.x, record_only=True)
        return ra, rb

    def backward(self, tape, dy, want_dx=False, param_grads=True, on_unit_done=None, addend=None):
        """on_unit_done(name), addend: as Generator.backward -- both networks of the pair finish a layer together."""
        nb = len(self.blocks)
        done = on_unit_done if on_unit_done is not None else (lambda name: None)
        d = dy
        for u, r in zip(reversed(self.tail), (tape[5 + nb], tape[4 + nb], tape[3 + nb])):
            d = u.backward(r, d, True, param_grads)
            done(u.ua.name)
        for k in range(nb - 1, -1, -1):
            ua, ub = self.blocks[k]
            ra, rb = self._block_records(k, tape[3 + k])
            t = ub.backward(rb, d, True, param_grads)
            done(ub.ua.name)
            d = ua.backward(ra, t, True, param_grads, addend=d)      # + skip gradient (fused into the data-gradient epilogue)
            done(ua.ua.name)
        d = self.head[2].backward(tape[2], d, True, param_grads)
        done("c3")
        d = self.head[1].backward(tape[1], d, True, param_grads)
        done("c2")
        d = self.head[0].backward(tape[0], d, want_dx, param_grads, addend=addend)
        done("c1")
        return d


class DiscriminatorPair:
    """discriminator (module.py:272-318) of two discriminators in lockstep on stacked images and masks."""

    def __init__(self, da, db):
        self.a, self.b = da, db
        self.units = [_PairUnit(ua, ub) for ua, ub in zip(da.units, db.units)]
        self.h4 = _PairUnit(da.h4, db.h4)

    def forward(self, x, mask):
        tape, h = [], x
        for u in self.units:
            h, r = u.forward(h)
            tape.append(r)
        h4, r = self.h4.forward(h)
        tape.append(r)
        out = K.mask_reduce_fwd(h4, mask, self.a.segment_class)                    # module.py:312-314
        tape.append((mask, tuple(h4.shape)))
        return out, tape

    def backward(self, tape, dlogits, want_dx=False, param_grads=True, addend=None):
        mask, h4_shape = tape[-1]
        d = K.mask_reduce_bwd(dlogits.contiguous(), mask, h4_shape, self.a.dtype, self.a.segment_class)
        d = self.h4.backward(tape[-2], d, True, param_grads)
        for i in range(len(self.units) - 1, -1, -1):
            d = self.units[i].backward(tape[i], d, want_dx or i > 0, param_grads, addend=addend if i == 0 else None)
        return d

    def slice_tape(self, tape, lo, hi):
        """The forward records of images [lo, hi) of a stacked pass, as views: a backward pass through PART of the batch (the
        cycle step sends reals and fakes through the discriminators as one pass, the generators' loss only needs the fakes).
        The slice must hold as many images of the first network as of the second."""
        k = (hi - lo) // 2
        out = []
        for pu, (g, x, xc, stats) in zip(self.units + [self.h4], tape[:-1]):
            xs = x[lo:hi]
            out.append((pu.ua.geom(xs[:k]), xs, xc[lo:hi], None if stats is None else stats[lo:hi]))
        mask, h4_shape = tape[-1]
        out.append((mask[lo:hi], (hi - lo,) + tuple(h4_shape[1:])))
        return out


# ----------------------------------------------------------------------------- autograd facade
class _NetFn(torch.autograd.Function):
    """Lets ``generator(x)`` / ``discriminator([x, mask])`` be used under torch autograd (tape-style
    callers like model.py:170-197).  The training loop of ``sggan`` does not go through this."""

    @staticmethod
    def forward(ctx, net, mask, x, *params):
        y, tape = net._run(x) if mask is None else net._run(x, mask)
        ctx.net, ctx.tape = net, tape
        return y

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        gbuf = torch.zeros_like(net.P.grad)
        dx = net._run_backward(ctx.tape, dy.to(torch.float32), gbuf, ctx.needs_input_grad[2])
        grads = tuple(net.P.g(n, buf=gbuf) for n in net.P.names())
        return (None, None, dx) + grads


def _apply_net(net, x, mask=None):
    x = torch.as_tensor(x)
    if x.device != net.device:
        x = x.to(net.device)
    params = net.trainable_variables
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        return _NetFn.apply(net, mask, x, *params)
    y, _ = net._run(x) if mask is None else net._run(x, mask)
    return y


def generator_resnet(**kw) -> Generator:
    """module.py:219 -- returns a callable ``G(x)``; keyword dims default to the reference constants."""
    return Generator(**kw)


def discriminator(**kw) -> Discriminator:
    """module.py:272 -- returns a callable ``D([x, mask])``."""
    return Discriminator(**kw)
