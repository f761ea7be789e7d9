"""``sggan`` -- host-side mirror of the reference's trainer object (model.py:39-566), hot path only.

``train_step`` follows model.py:169-200 line by line (documented deviations: D2 -- ``fake_A =
G(real_A)`` every step and ``da_fake`` evaluated once, since :187 and :188 are the same values).
Everything numeric runs in libsggan.so; this file sequences kernels, owns the flat
parameter/gradient buffers and (optionally) overlaps the data-parallel gradient all-reduce
(RCCL over xGMI; one bucket per network) with the remaining backward work.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from . import _abi as A
from . import kernels as K
from .graph import StepProgram
from .module import Adam, Discriminator, DiscriminatorPair, Generator, GeneratorPair
from .utils import ImagePool, StaticImagePool

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32,
           "float32": torch.float32, torch.bfloat16: torch.bfloat16, torch.float32: torch.float32}


def default_args(**over):
    """The reference's argparse defaults (main.py:14-43) for the flags the step consumes, plus the
    build's own knobs (dtype/device/n_blocks/seed), as a namespace."""
    a = dict(batch_size=1, image_height=128, image_width=128, input_nc=3, output_nc=3, ngf=64, ndf=64,
             segment_class=34, beta1=0.5, lr=0.0002, L1_lambda=10.0, Lg_lambda=5.0, use_resnet=True, use_pix2pix=False,
             use_lsgan=True, ratio_gan2seg=10, max_size=50, phase="train", dataset_dir="city",
             dtype="bf16", device="cuda", n_blocks=9, seed=19, graph=False, mixed=False, paired=True,
             fuse_in_stats=True, fuse_in_bwd=False, g_buckets=3, keep_tapes=False, group2=True, d_quad=True,
             checkpoint_blocks=False, use_pool=False, pool_static=False, fuse_in_stats_deconv=False, fuse_in_stats_stem=False)
    a.update(over)
    return SimpleNamespace(**a)


class sggan(object):
    # model.py:151 ``LAMBDA = 100`` (hard-coded; --L1_lambda is ignored by the live step) and
    # model.py:205 ``lr = 0.001`` (hard-coded; --lr is ignored) -- SURVEY.md 5 "Config / flags".
    LAMBDA = 100.0
    LR = 0.001

    def __init__(self, args=None, **kw):
        args = args if args is not None else default_args(**kw)
        g = lambda k, d=None: getattr(args, k, d)
        self.batch_size = g("batch_size", 1)
        self.image_width, self.image_height = g("image_width", 128), g("image_height", 128)
        self.input_c_dim, self.output_c_dim = g("input_nc", 3), g("output_nc", 3)
        self.segment_class = g("segment_class", 34)
        self.use_pix2pix = bool(g("use_pix2pix", False))
        if self.use_pix2pix or not g("use_resnet", True):
            raise NotImplementedError("only the ResNet generator / mask discriminator path is built "
                                      "(generator_unet / *_pix2pix are out of scope, SURVEY.md 2.1)")
        self.dtype = _DTYPES[g("dtype", "bf16")]
        self.device = torch.device(g("device", "cuda"))
        seed = g("seed", 19)
        self.discriminator = Discriminator(df_dim=g("ndf", 64), in_c=self.output_c_dim, segment_class=self.segment_class,
                                           dtype=self.dtype, device=self.device, seed=seed + 1)          # model.py:54
        self.generator = Generator(gf_dim=g("ngf", 64), in_c=self.input_c_dim, out_c=self.output_c_dim,
                                   n_blocks=g("n_blocks", 9), dtype=self.dtype, device=self.device, seed=seed)  # :56
        self.beta1 = g("beta1", 0.5)
        self.lr = self.LR
        # cycle mode (north_star unit, deviation D5): G_A->B = self.generator, D_A = self.discriminator, plus G_B->A and D_B
        self.cycle = bool(g("cycle", False))
        self.L1_lambda, self.Lg_lambda = float(g("L1_lambda", 10.0)), float(g("Lg_lambda", 5.0))
        self.use_lsgan = bool(g("use_lsgan", True))
        self.cycle_lr = float(g("lr", 0.0002))
        if self.cycle:
            self.generator_BA = Generator(gf_dim=g("ngf", 64), in_c=self.output_c_dim, out_c=self.input_c_dim,
                                          n_blocks=g("n_blocks", 9), dtype=self.dtype, device=self.device, seed=seed + 2)
            self.discriminator_B = Discriminator(df_dim=g("ndf", 64), in_c=self.output_c_dim, segment_class=self.segment_class,
                                                 dtype=self.dtype, device=self.device, seed=seed + 3)
            self.real_B = self.seg_B = self.mask_B = self.fake_B = None
        # step I/O attributes (model.py:250-256 sets the inputs; :260 reads the losses)
        self.real_A = self.seg_A = self.mask_A = self.fake_A = None
        self._loss = torch.zeros(2, dtype=torch.float32, device=self.device)    # [gen_loss, disc_loss] on device
        self.gen_loss, self.disc_loss = self._loss[0:1], self._loss[1:2]
        self._dp = None
        self._world = 1
        self.dataset_dir = g("dataset_dir", "city")
        # model.py:79 builds ImagePool(args.max_size) but the fork never calls it; it is used by the cycle step when
        # use_pool is set (upstream SG-GAN behaviour: D sees a history of fakes)
        self.use_pool = bool(g("use_pool", False))
        # pool_static: the pool step with a fixed launch sequence (utils.StaticImagePool: decisions on the host, selects on the
        # device) -- what graph replay needs; the dynamic form (D judges the pooled fakes in their own pass only when the pool
        # hands back older ones) stays the default of the eager path
        self.pool_static = bool(g("pool_static", False)) or (self.use_pool and bool(g("graph", False)))
        self.pool = (StaticImagePool if self.pool_static else ImagePool)(g("max_size", 50), rng=g("pool_rng", None))
        self.pair_wgrads = bool(g("pair_wgrads", True))   # cycle step: one weight-gradient launch per layer for both applications of a G
        self.batch_d_real_fake = bool(g("batch_d_real_fake", True))   # pool mode: D(real) and D(pooled fakes) as one 2N pass
        self.d_quad = bool(g("d_quad", True))             # reals and fakes through the discriminator(s) as ONE stacked pass (both step flavours)
        self.gen_loss_metric, self.disc_loss_metric, self._metric_n = 0.0, 0.0, 0
        # model.py:83-84 / 205-207: one Keras Adam per network (lr hard-coded 1e-3 on the live step; the cycle step
        # uses --lr).  d_optim / g_optim are the reference's attribute names; the cycle step adds the other two.
        lr = self.cycle_lr if self.cycle else self.lr
        self.g_optim = Adam(self.generator, lr, self.beta1)
        self.d_optim = Adam(self.discriminator, lr, self.beta1)
        if self.cycle:
            self.g_optim_BA = Adam(self.generator_BA, lr, self.beta1)
            self.d_optim_B = Adam(self.discriminator_B, lr, self.beta1)
        # cycle step: run the two generators (and the two discriminators) in lockstep on stacked batches (module._PairUnit);
        # bit-identical to the one-network-at-a-time sequencing, which stays for the image pool and for mixed mode
        self.paired = bool(g("paired", True))
        # mixed precision: bf16 storage, but the activation-gradient chain through the generators' residual blocks in f32
        self.mixed = bool(g("mixed", False)) and self.dtype == torch.bfloat16
        for net in self.networks():
            net.mixed = self.mixed
            # conv epilogue -> norm statistics (on), data-gradient epilogue -> norm-backward sums (opt-in; module.py)
            net.fuse_in_stats, net.fuse_in_bwd = bool(g("fuse_in_stats", True)), bool(g("fuse_in_bwd", False))
            net.fuse_in_stats_deconv, net.fuse_in_stats_stem = bool(g("fuse_in_stats_deconv", False)), bool(g("fuse_in_stats_stem", False))
            net.group2 = bool(g("group2", True))
            # activation checkpointing (BASELINE.json configs[4]): the generators keep each residual block's input only and
            # re-run the block (module.py:208-217) in backward -- bitwise the same gradients, + 2 conv forwards per block
            net.checkpoint_blocks = bool(g("checkpoint_blocks", False)) and isinstance(net, Generator)
        # data parallel: each generator's gradient buffer is exchanged as this many contiguous layer-group buckets, launched
        # in backward-completion order so that all but the last overlap the rest of the backward pass
        self.g_buckets = max(1, int(g("g_buckets", 3)))
        # diagnostics: keep the last step's forward records (saved activations) reachable as ``self.tapes`` -- the parity tests
        # read from them which side of each ReLU / LeakyReLU kink the kernels took (tests/kink_helpers.py).  Off: they are dropped
        # when the step returns, so their memory is reused at once.
        self.keep_tapes = bool(g("keep_tapes", False))
        self.tapes = None
        # HIP-graph replay of the step (graph.py): recorded at the first train_step after enable_graph()
        self.use_graph = bool(g("graph", False))
        self._program = None
        self._static_in = {}
        self._stack_bufs = {}                       # persistent stacked-batch buffers of the step (_stacked)

    # ------------------------------------------------------------------ data parallel (new capability)
    def enable_data_parallel(self, process_group=None):
        """Average gradients over ranks with one all-reduce per network, launched as soon as that
        network's backward has been queued so it overlaps with the rest of the step (dp.GradExchange)."""
        from .dp import GradExchange
        self._dp = GradExchange(process_group)
        self._world = self._dp.world
        for net in self.networks():                             # identical replicas: broadcast rank 0's parameters
            self._dp.broadcast_(net.P.flat)
            net.P.version += 1
        self._program = None                                    # a recorded step has no collectives in it: record again
        return self

    def _host(self, fn):
        """A host-side action at this point of the step (graph.StepProgram.host): called right away on the eager
        path; while the step is being recorded it ends the current HIP-graph segment and is replayed between segments."""
        return K.host(fn)

    def _allreduce(self, net, lo=None, hi=None, handle=None):
        """Launch the all-reduce of one network's gradient bucket (the whole flat buffer, or its [lo, hi) range); the
        returned handle's wait() orders the stream behind every bucket launched on it."""
        if self._dp is None:
            return None
        h = handle if handle is not None else _PendingReduce(self)
        buf = net.P.grad if lo is None else net.P.grad[lo:hi]
        def launch():
            h.works.append(self._dp.allreduce_async(buf))
        self._host(launch)
        return h

    def _bucketed_allreduce(self, nets):
        """(on_unit_done, handles) for the LAST backward pass of the generator(s) ``nets`` this step: each network's flat
        gradient buffer is cut into ``g_buckets`` contiguous layer groups (module.bucket_plan); a group's all-reduce is launched
        as soon as the backward of its first layer -- weight gradient included -- has been queued, i.e. while the layers in front
        of it are still being differentiated.  Networks that run in lockstep (the paired cycle step) share the host action."""
        if self._dp is None:
            return None, [None] * len(nets)
        plans = [{first: (lo, hi) for first, lo, hi in net.bucket_plan(self.g_buckets)} for net in nets]
        handles = [_PendingReduce(self) for _ in nets]

        def on_unit_done(name):
            ready = [(net, h, plan[name]) for net, h, plan in zip(nets, handles, plans) if name in plan]
            if not ready:
                return
            # a bucket may only travel once every weight gradient inside its range has been launched: a deferred (paired)
            # weight gradient that ran after its range's all-reduce would make the ranks diverge silently
            for net, _, (lo, hi) in ready:
                pend = [u for u in net.pending_wgrads() if lo <= net.P.index[u + "_w"][0] < hi]
                assert not pend, f"bucket [{lo}, {hi}) launched at {name} with weight gradients still deferred: {pend}"
            bufs = [(h, net.P.grad[lo:hi]) for net, h, (lo, hi) in ready]
            def launch():
                for h, buf in bufs:
                    h.works.append(self._dp.allreduce_async(buf))
            self._host(launch)
        return on_unit_done, handles

    # ------------------------------------------------------------------ HIP-graph replay (graph.py)
    def enable_graph(self, flag=True):
        """Replay the step from captured HIP graphs instead of dispatching ~1 250 launches from Python each step."""
        self.use_graph = bool(flag)
        if not flag:
            self._program = None
        return self

    _INPUTS_REF = ("real_A", "seg_A", "mask_A")
    _INPUTS_CYCLE = ("real_A", "seg_A", "mask_A", "real_B", "seg_B", "mask_B")

    def _convert_input(self, name, x):
        if name.startswith("mask"):
            m = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32))
            return m.to(device=self.device, dtype=torch.float32).contiguous()
        return self._prep(x)

    def _stage_inputs(self):
        """Step inputs -> the static device buffers the recorded graphs read.  A buffer the caller fills in place (it is
        what ``self.real_A`` etc. point at after the first graph step) costs nothing; anything else is converted and copied."""
        names = self._INPUTS_CYCLE if self.cycle else self._INPUTS_REF
        fresh = False
        for n in names:
            x = getattr(self, n)
            buf = self._static_in.get(n)
            if buf is not None and x is buf:
                continue
            t = self._convert_input(n, x)
            if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                self._static_in[n] = t.clone()
                fresh = True
            else:
                buf.copy_(t)
            setattr(self, n, self._static_in[n])
        return fresh

    def _graph_step(self):
        if self.use_pool and not self.pool_static:
            raise NotImplementedError("graph replay with the dynamic image pool: its random swaps change the step's launch "
                                      "sequence; build the model with pool_static=True (or graph=True) or run the pool step eagerly")
        if self._stage_inputs():
            self._program = None                   # new shapes: record again
        if self._program is None:
            self._record()
        if self.use_pool:
            self.pool.stage(self.device)           # this step's pool decisions -> the device tensor the recorded selects read
        self._program.replay()
        for n in self.networks():
            n.P.version += 1                       # the replayed Adam changed the parameters: eager callers must re-pack
        return self.gen_loss, self.disc_loss

    def _record(self):
        """Warm up eagerly (kernel attributes, workspaces, weight-pack tables), put the parameters and optimizer state
        back, then record ONE step without executing it."""
        nets = self.networks()
        keep = [(n.P.flat.clone(), n.P.m.clone(), n.P.v.clone(), n.P.iterations.clone()) for n in nets]
        loss_keep = self._loss.clone()
        hook, K.PROFILE = K.PROFILE, None          # timing hooks (bench.py) record events: not while recording
        try:
            if self.use_pool:                      # the warm-up step: "return the input, store nothing" (no decision is consumed)
                cur = max(self.pool.maxsize, 0)
                if self.pool.ctrl is None:
                    self.pool.ctrl = torch.zeros(4, dtype=torch.int64, device=self.device)
                self.pool.ctrl.fill_(cur)
            self._step_body()
            torch.cuda.synchronize(self.device)
            for n, (f, m, v, it) in zip(nets, keep):
                n.P.flat.copy_(f); n.P.m.copy_(m); n.P.v.copy_(v); n.P.iterations.copy_(it)
                n.P.version += 1                   # the recorded step starts by re-packing the conv weights
            self._loss.copy_(loss_keep)
            prog = StepProgram(self.device)
            K._RECORDER = prog
            try:
                prog.record(self._step_body)
            finally:
                K._RECORDER = None
            # addresses baked into the recorded launches that do not come from the capture pool: the shared scratch workspace
            # and the networks' scratch vectors (allocated by the eager warm-up).  The program keeps them alive, so a later,
            # larger workspace request elsewhere in the process cannot hand their memory to another tensor under a replay.
            prog.keep = K.workspace_refs() + [n._scratch for n in nets if n._scratch is not None]
            self._program = prog
        finally:
            K.PROFILE = hook

    # ------------------------------------------------------------------ the hot path
    def _prep(self, x, out=None):
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32))
        return self.generator.to_internal(t.to(self.device), out=out)

    def _stacked(self, key, shape, dtype):
        """A persistent device buffer for one of the step's stacked batches ([real_A; real_B], the discriminators' stacked input
        and masks): inputs are converted straight into its slices and the generators' last layer writes the fakes into it, so
        the step concatenates nothing."""
        bufs = self._stack_bufs
        buf = bufs.get(key)
        if buf is None or tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
            buf = bufs[key] = torch.empty(tuple(shape), dtype=dtype, device=self.device)
        return buf

    def train_step(self, args=None):
        """model.py:169-200.  Reads ``real_A, seg_A`` (N,H,W,3) in [0,1] and ``mask_A`` (N,mh,mw,C);
        writes ``fake_A, gen_loss, disc_loss``; updates G, D and both Adam states."""
        if self.use_graph:
            return self._graph_step()
        if self.use_pool and self.pool_static:
            self.pool.stage(self.device)
        return self._step_body()

    def _step_body(self):
        if self.cycle:
            return self._train_step_cycle()
        G, D = self.generator, self.discriminator
        # :186-188.  d_quad: D sees [seg; fake] as ONE stacked pass (half the launches; its tail layers are launch-latency bound);
        # the generator's loss backpropagates through the fake slice of the same tape.  Per image the same arithmetic.  The
        # stacked input is one persistent buffer: seg is converted into its first half, G writes the fakes into the second.
        stack = self.d_quad
        real = self._prep(self.real_A)
        N = real.shape[0]
        mask = self._convert_input("mask_A", self.mask_A)
        if stack:
            d_in = self._stacked("d_in", (2 * N,) + tuple(real.shape[1:]), real.dtype)
            seg = self._prep(self.seg_A, out=d_in[:N])
            mask2 = torch.cat([mask, mask], out=self._stacked("d_mask", (2 * N,) + tuple(mask.shape[1:]), mask.dtype))
        else:
            seg = self._prep(self.seg_A)
        G.P.zero_grad()
        D.P.zero_grad()

        fake, tG = G.forward(real, out=d_in[N:] if stack else None)         # :175-179 (D2)
        if stack:
            da_both, tDq = D.forward(d_in, mask2)
            da_real, da_fake = da_both[:N], da_both[N:]
            tDr, tDf = None, D.slice_tape(tDq, N, 2 * N)
        else:
            da_real, tDr = D.forward(seg, mask)                             # :186
            da_fake, tDf = D.forward(fake, mask)                            # :187 (= :188)

        # losses + their logit gradients, all on device (no host sync)
        gl, dl = self._loss[0:1], self._loss[1:2]
        d_fake_g = torch.empty_like(da_fake)
        d_both_d = torch.empty((2 * N,) + tuple(da_real.shape[1:]), dtype=da_real.dtype, device=da_real.device)
        d_real_d, d_fake_d = d_both_d[:N], d_both_d[N:]
        dfake_l1 = torch.empty_like(fake)
        K.bce_logits(da_fake, 1.0, gl, d_fake_g)                            # :153 gan_loss
        K.l1_loss(seg, fake, self.output_c_dim, gl, dfake_l1, weight=self.LAMBDA, accumulate=True)   # :155-156
        K.bce_logits(da_real, 1.0, dl, d_real_d)                            # :162
        K.bce_logits(da_fake, 0.0, dl, d_fake_d, accumulate_loss=True)      # :163-164 (adds to disc_loss)

        # disc_tape.gradient(disc_loss, D vars)  (:197)
        if stack:
            D.backward(tDq, d_both_d, want_dx=False, param_grads=True)
        else:
            D.backward(tDr, d_real_d, want_dx=False, param_grads=True)
            D.backward(tDf, d_fake_d, want_dx=False, param_grads=True)
        hD = self._allreduce(D)
        # gen_tape.gradient(gen_loss, G vars)    (:196): through D's data path, then G
        dfake = D.backward(tDf, d_fake_g, want_dx=True, param_grads=False, addend=dfake_l1)   # (the join rides on h0's data-gradient store)
        hook, (hG,) = self._bucketed_allreduce((G,))
        G.backward(tG, dfake, want_dx=False, param_grads=True, on_unit_done=hook)

        scale = 1.0 / self._world
        if hD is not None:
            hD.wait()
        self.d_optim.apply_gradients(grad_scale=scale)                     # :200
        if hG is not None:
            hG.wait()
        self.g_optim.apply_gradients(grad_scale=scale)                     # :199
        self._fake_internal = fake
        self.fake_A = _LazyUnpad(fake, self.output_c_dim)
        self.da_real, self.da_fake = da_real, da_fake
        if self.keep_tapes:            # (stacked pass: views of its records, images [0, N) = D(seg), [N, 2N) = D(fake))
            self.tapes = {"G": tG, "D_real": D.slice_tape(tDq, 0, N) if stack else tDr, "D_fake": tDf}
        return self.gen_loss, self.disc_loss

    def networks(self):
        return (self.generator, self.discriminator) + ((self.generator_BA, self.discriminator_B) if self.cycle else ())

    def _train_step_cycle(self):
        """2G+2D step assembled from the reference's defined-not-wired criteria (deviation D5):
        generator_loss / discriminator_loss (model.py:114-133) with criterionGAN = mae_criterion (--use_lsgan) or
        sce_criterion, abs_criterion cycle terms x --L1_lambda, gradloss_criterion x --Lg_lambda weighted by the
        segmentation-edge indicator (model.py:108-119).  Fakes in domain B are judged on A's mask and vice versa."""
        if self.paired and not self.use_pool and not self.mixed:
            return self._train_step_cycle_paired()
        Gab, Gba, Da, Db = self.generator, self.generator_BA, self.discriminator, self.discriminator_B
        prep_mask = lambda m: (m if isinstance(m, torch.Tensor) else torch.as_tensor(np.asarray(m, dtype=np.float32))).to(
            device=self.device, dtype=torch.float32).contiguous()
        rA, rB = self._prep(self.real_A), self._prep(self.real_B)
        sA, sB = self._prep(self.seg_A), self._prep(self.seg_B)
        mA, mB = prep_mask(self.mask_A), prep_mask(self.mask_B)
        for net in (Gab, Gba, Da, Db):
            net.P.zero_grad()
        C = self.output_c_dim

        fake_B, t1 = Gab.forward(rA)
        cyc_A, t4 = Gba.forward(fake_B)
        fake_A, t3 = Gba.forward(rB)
        cyc_B, t2 = Gab.forward(fake_A)
        DB_fake, tDBf = Db.forward(fake_B, mA)
        DA_fake, tDAf = Da.forward(fake_A, mB)
        wA, wB = K.seg_edge_weight(sA, C), K.seg_edge_weight(sB, C)

        crit = K.mse_const if self.use_lsgan else K.bce_logits
        gl, dl = self._loss[0:1], self._loss[1:2]
        e = torch.empty_like
        gA_g, gB_g = e(DA_fake), e(DB_fake)                  # d g_loss / d logits
        crit(DA_fake, 1.0, gl, gA_g)
        crit(DB_fake, 1.0, gl, gB_g, accumulate_loss=True)
        d_cycA, d_cycB = e(cyc_A), e(cyc_B)
        K.l1_loss(rA, cyc_A, C, gl, d_cycA, weight=self.L1_lambda, accumulate=True)
        K.l1_loss(rB, cyc_B, C, gl, d_cycB, weight=self.L1_lambda, accumulate=True)
        d_fA, d_fB = e(fake_A), e(fake_B)                    # gradient-sensitive terms write, the rest accumulate
        K.gradloss(fake_A, rB, wB, C, gl, d_fA, lam=self.Lg_lambda, accumulate_loss=True)
        K.gradloss(fake_B, rA, wA, C, gl, d_fB, lam=self.Lg_lambda, accumulate_loss=True)
        # the discriminators judge a HISTORY of fakes when the image pool is on (utils.py:27-53); the pool returns the
        # current fakes until it is full, and then (p = 1/2) older ones, which need their own D forward
        pooled_A = pooled_B = False
        pfA = pfB = smA = smB = None
        if self.use_pool:
            pfA, pfB, smB, smA = self.pool([fake_A, fake_B, mB, mA])       # fake_A is judged on mask_B, fake_B on mask_A
            pooled_A, pooled_B = pfA is not fake_A, pfB is not fake_B      # (static pool: always fresh tensors -> always the stacked pass)
            assert not self.pool_static or self.batch_d_real_fake
        # D's two loss terms.  When the pool hands back older fakes they need their own D forward anyway: it is run
        # together with the real batch as ONE pass over 2N images (instance norm is per image, so this is exact) --
        # half the launches, and D's small tail layers (5x13 ... 1x5 maps) fill more of the chip.
        def d_loss_pass(Dn, real, m_real, DR, tR, pooled, fakes, m_fakes, DF, tF, first):
            if pooled and self.batch_d_real_fake:
                out, tape = Dn.forward(torch.cat([real, fakes]), torch.cat([m_real, m_fakes]))
                n = real.shape[0]
                grad = e(out)
                crit(out[:n], 1.0, dl, grad[:n], weight=0.5, accumulate_loss=not first)
                crit(out[n:], 0.0, dl, grad[n:], weight=0.5, accumulate_loss=True)
                Dn.backward(tape, grad)
                return
            if pooled:
                DF, tF = Dn.forward(fakes, m_fakes)
            g_r, g_f = e(DR), e(DF)
            crit(DR, 1.0, dl, g_r, weight=0.5, accumulate_loss=not first)
            crit(DF, 0.0, dl, g_f, weight=0.5, accumulate_loss=True)
            Dn.backward(tR, g_r); Dn.backward(tF, g_f)

        # discriminator gradients (fakes are constants here)
        if pooled_A and self.batch_d_real_fake:
            d_loss_pass(Da, rA, mA, None, None, True, pfA, smB, None, None, True)
        else:
            DA_real, tDAr = Da.forward(rA, mA)
            d_loss_pass(Da, rA, mA, DA_real, tDAr, pooled_A, pfA if pooled_A else None, smB if pooled_A else None, DA_fake, tDAf, True)
        hDa = self._allreduce(Da)
        if pooled_B and self.batch_d_real_fake:
            d_loss_pass(Db, rB, mB, None, None, True, pfB, smA, None, None, False)
        else:
            DB_real, tDBr = Db.forward(rB, mB)
            d_loss_pass(Db, rB, mB, DB_real, tDBr, pooled_B, pfB if pooled_B else None, smA if pooled_B else None, DB_fake, tDBf, False)
        hDb = self._allreduce(Db)
        # generator gradients: cycle terms first (they reach the other generator through the fakes).  Each generator is
        # applied twice, so the weight gradients of its 3x3 layers are paired: deferred here, launched with the second pass
        Gab.pair_wgrads = Gba.pair_wgrads = self.pair_wgrads
        # (the gradient joins ride on the first layer's data-gradient store -- ``addend`` -- instead of separate passes)
        d_fB = Gba.backward(t4, d_cycA, want_dx=True, addend=d_fB)
        d_fA = Gab.backward(t2, d_cycB, want_dx=True, addend=d_fA)
        d_fB = Db.backward(tDBf, gB_g, want_dx=True, param_grads=False, addend=d_fB)
        d_fA = Da.backward(tDAf, gA_g, want_dx=True, param_grads=False, addend=d_fA)
        hook, (hGba,) = self._bucketed_allreduce((Gba,))
        Gba.backward(t3, d_fA, on_unit_done=hook)               # second application: the paired weight gradients run here
        Gba.flush_wgrads(); Gba.pair_wgrads = False             # (nothing is left to flush unless a partner never came)
        hook, (hGab,) = self._bucketed_allreduce((Gab,))
        Gab.backward(t1, d_fB, on_unit_done=hook)
        Gab.flush_wgrads(); Gab.pair_wgrads = False

        scale = 1.0 / self._world
        for opt, h in ((self.d_optim, hDa), (self.d_optim_B, hDb), (self.g_optim_BA, hGba), (self.g_optim, hGab)):
            if h is not None:
                h.wait()
            opt.apply_gradients(grad_scale=scale)
        self.fake_A, self.fake_B = _LazyUnpad(fake_A, self.input_c_dim), _LazyUnpad(fake_B, C)
        self.cyc_A, self.cyc_B = _LazyUnpad(cyc_A, self.input_c_dim), _LazyUnpad(cyc_B, C)
        return self.gen_loss, self.disc_loss

    def _train_step_cycle_paired(self):
        """The cycle step with the symmetric halves of the objective in lockstep: every activation tensor stacks the two
        translation directions on the batch dimension -- [real_A; real_B] -> [fake_B; fake_A] -> [cyc_A; cyc_B] through
        (G_A->B, G_B->A) then (G_B->A, G_A->B); the fakes through (D_B, D_A), the reals through (D_A, D_B).  Convolutions still
        run per network; instance norms, activations and gradient joins run once per pair over twice the bytes.  Same kernels
        per image as _train_step_cycle: with ``d_quad=False`` losses, images and data gradients are bit-identical to it; weight
        gradients of layers whose two networks share a launch (sgg_conv2d_bwd_weight_pair2 and the all-taps / LDS-DMA shapes of
        sgg_conv2d_bwd_weight_group2: half as many split slabs per network) are equal up to f32 summation order (held to 1e-5
        of the tensor norm by the tests).  With ``d_quad`` (the default) the discriminators see reals and fakes as one 4N pass,
        whose small layers take other split-K plans than two 2N passes: the same mathematics in another f32 summation order
        (both forms are held to the float64 restatement at 2e-4 on every gradient tensor, tests/test_gpu_step.py)."""
        Gab, Gba, Da, Db = self.generator, self.generator_BA, self.discriminator, self.discriminator_B
        if getattr(self, "_pairs", None) is None:
            self._pairs = (GeneratorPair(Gab, Gba), GeneratorPair(Gba, Gab), DiscriminatorPair(Db, Da), DiscriminatorPair(Da, Db))
        Gp1, Gp2, Dpf, Dpr = self._pairs
        # Discriminators.  d_quad: reals AND fakes go through (D_B, D_A) as ONE stacked pass [real_B; fake_B | fake_A; real_A] --
        # half the launches of the two passes it replaces (the discriminators' tails are launch-latency bound) and the weight
        # gradients of both applications in one launch; the generators' loss backpropagates through the middle (fake) slice.
        # The stacked batches are persistent buffers: the inputs are converted straight into their slices and the generators'
        # last layer writes [fake_B; fake_A] into the middle of the discriminators' input -- nothing is concatenated.
        quad = self.d_quad
        shp = tuple(self.real_A.shape)
        n = shp[0]
        lo, hi = slice(0, n), slice(n, 2 * n)
        act_shape = tuple(shp[1:3]) + (K.cpad(self.input_c_dim),)
        x1 = self._stacked("x1", (2 * n,) + act_shape, self.dtype)
        rA, rB = self._prep(self.real_A, out=x1[lo]), self._prep(self.real_B, out=x1[hi])
        sA, sB = self._prep(self.seg_A), self._prep(self.seg_B)
        mA, mB = self._convert_input("mask_A", self.mask_A), self._convert_input("mask_B", self.mask_B)
        dq = None
        if quad:
            dq = self._stacked("dq", (4 * n,) + act_shape, self.dtype)
            self._prep(self.real_B, out=dq[:n]); self._prep(self.real_A, out=dq[3 * n:])
            mq = torch.cat([mB, mA, mB, mA], out=self._stacked("mq", (4 * n,) + tuple(mA.shape[1:]), mA.dtype))
            m_ab = mq[n:3 * n]                                # [mask_A; mask_B]
        else:
            m_ab = torch.cat([mA, mB])
        for net in (Gab, Gba, Da, Db):
            net.P.zero_grad()
        C = self.output_c_dim
        f1, t1 = Gp1.forward(x1, out=dq[n:3 * n] if quad else None)   # [fake_B; fake_A]
        c2, t2 = Gp2.forward(f1)                              # [cyc_A; cyc_B]
        if quad:
            Dq, tDq = Dpf.forward(dq, mq)
            Df, tDf = Dq[n:3 * n], Dpf.slice_tape(tDq, n, 3 * n)
        else:
            Df, tDf = Dpf.forward(f1, m_ab)                   # [D_B(fake_B | mask_A); D_A(fake_A | mask_B)]
        wA, wB = K.seg_edge_weight(sA, C), K.seg_edge_weight(sB, C)

        crit = K.mse_const if self.use_lsgan else K.bce_logits
        gl, dl = self._loss[0:1], self._loss[1:2]
        e = torch.empty_like
        g_g = e(Df)                                           # d g_loss / d logits, stacked like Df
        crit(Df[hi], 1.0, gl, g_g[hi])                        # D_A(fake_A)   (the loss terms in _train_step_cycle's order)
        crit(Df[lo], 1.0, gl, g_g[lo], accumulate_loss=True)  # D_B(fake_B)
        d_cyc = e(c2)
        K.l1_loss(rA, c2[lo], C, gl, d_cyc[lo], weight=self.L1_lambda, accumulate=True)
        K.l1_loss(rB, c2[hi], C, gl, d_cyc[hi], weight=self.L1_lambda, accumulate=True)
        d_f = e(f1)                                           # gradient-sensitive terms write, the rest accumulate
        K.gradloss(f1[hi], rB, wB, C, gl, d_f[hi], lam=self.Lg_lambda, accumulate_loss=True)
        K.gradloss(f1[lo], rA, wA, C, gl, d_f[lo], lam=self.Lg_lambda, accumulate_loss=True)

        # discriminator gradients (fakes are constants here): reals through (D_A, D_B), fakes through (D_B, D_A)
        tDr = None
        if quad:
            g_q = e(Dq)                                       # [real_B; fake_B | fake_A; real_A], the loss terms in the same order as below
            crit(Dq[3 * n:], 1.0, dl, g_q[3 * n:], weight=0.5, accumulate_loss=False)
            crit(Dq[2 * n:3 * n], 0.0, dl, g_q[2 * n:3 * n], weight=0.5, accumulate_loss=True)
            crit(Dq[:n], 1.0, dl, g_q[:n], weight=0.5, accumulate_loss=True)
            crit(Dq[n:2 * n], 0.0, dl, g_q[n:2 * n], weight=0.5, accumulate_loss=True)
            Dpf.backward(tDq, g_q)
        else:
            Dr, tDr = Dpr.forward(x1, m_ab)                   # [D_A(real_A | mask_A); D_B(real_B | mask_B)]
            g_r, g_fk = e(Dr), e(Df)
            crit(Dr[lo], 1.0, dl, g_r[lo], weight=0.5, accumulate_loss=False)
            crit(Df[hi], 0.0, dl, g_fk[hi], weight=0.5, accumulate_loss=True)
            crit(Dr[hi], 1.0, dl, g_r[hi], weight=0.5, accumulate_loss=True)
            crit(Df[lo], 0.0, dl, g_fk[lo], weight=0.5, accumulate_loss=True)
            Dpr.backward(tDr, g_r)
            Dpf.backward(tDf, g_fk)
        hDa, hDb = self._allreduce(Da), self._allreduce(Db)
        # generator gradients: cycle terms first (they reach the other generator through the fakes).  Each generator is applied
        # twice, so the weight gradients of its 3x3 layers are paired: deferred in the first pass, launched with the second
        Gab.pair_wgrads = Gba.pair_wgrads = self.pair_wgrads
        # (the gradient joins ride on the first layer's data-gradient store -- ``addend`` -- instead of separate passes)
        d_f = Gp2.backward(t2, d_cyc, want_dx=True, addend=d_f)
        d_f = Dpf.backward(tDf, g_g, want_dx=True, param_grads=False, addend=d_f)
        # second application of both generators: every deferred weight gradient runs inside this pass, and each layer group's
        # all-reduce is launched as soon as the group is complete -- only the last group's exchange has no backward work left
        # to hide behind (the four Adam launches run under it)
        hook, (hGab, hGba) = self._bucketed_allreduce((Gab, Gba))
        Gp1.backward(t1, d_f, on_unit_done=hook)
        assert self._dp is None or not (Gab.pending_wgrads() or Gba.pending_wgrads())   # (their buckets are already travelling)
        Gba.flush_wgrads(); Gab.flush_wgrads()                  # (nothing is left to flush unless a partner never came)
        Gab.pair_wgrads = Gba.pair_wgrads = False

        scale = 1.0 / self._world
        for opt, h in ((self.d_optim, hDa), (self.d_optim_B, hDb), (self.g_optim_BA, hGba), (self.g_optim, hGab)):
            if h is not None:
                h.wait()
            opt.apply_gradients(grad_scale=scale)
        self.fake_A, self.fake_B = _LazyUnpad(f1[hi], self.input_c_dim), _LazyUnpad(f1[lo], C)
        self.cyc_A, self.cyc_B = _LazyUnpad(c2[lo], self.input_c_dim), _LazyUnpad(c2[hi], C)
        if self.keep_tapes:
            # stacked [first network's images; second network's images] -- see the docstring.  d_quad: "D_quad" is the ONE
            # discriminator pass [D_B(real_B); D_B(fake_B) | D_A(fake_A); D_A(real_A)] and "D_fake" its middle slice (views);
            # otherwise "D_fake" = [D_B(fake_B); D_A(fake_A)] and "D_real" = [D_A(real_A); D_B(real_B)] are the two passes
            self.tapes = {"G_first": t1, "G_second": t2, "D_fake": tDf, "D_real": tDr, "D_quad": tDq if quad else None, "n": n}
        return self.gen_loss, self.disc_loss

    # ------------------------------------------------------------------ convenience
    def losses(self):
        """Host copies of (gen_loss, disc_loss) -- the two scalars model.py:260 prints."""
        v = self._loss.detach().cpu().tolist()
        return v[0], v[1]

    def state_dict(self):
        sd = {}
        names = ("G", "D", "G_BA", "D_B") if self.cycle else ("G", "D")
        for key, net in zip(names, self.networks()):
            P = net.P
            sd[key] = {"flat": P.flat.cpu(), "m": P.m.cpu(), "v": P.v.cpu(), "t": P.step_count}
        return sd

    def load_state_dict(self, sd):
        names = ("G", "D", "G_BA", "D_B") if self.cycle else ("G", "D")
        for key, net in zip(names, self.networks()):
            P = net.P
            P.flat.copy_(sd[key]["flat"]); P.m.copy_(sd[key]["m"]); P.v.copy_(sd[key]["v"])
            P.step_count = int(sd[key]["t"]); P.version += 1

    def _ckpt_paths(self, checkpoint_dir, ep=None):
        """model.py:454-456: <checkpoint_dir>/<dataset_dir>/{gen,disc}/cp-{epoch:04d}.ckpt"""
        import os
        base = os.path.join(checkpoint_dir, self.dataset_dir)
        f = (lambda sub: os.path.join(base, sub, "cp-%04d.ckpt" % ep)) if ep is not None else (lambda sub: os.path.join(base, sub))
        return f("gen"), f("disc")

    def save(self, checkpoint_dir, ep):
        """model.py:450-468 (weights per network under gen/ and disc/), plus what the reference omits: the Adam
        slots and step counts, so training resumes exactly (SURVEY.md 5 "Checkpoint / resume")."""
        import os
        gpath, dpath = self._ckpt_paths(checkpoint_dir, ep)
        sd = self.state_dict()
        for path, keys in ((gpath, ("G", "G_BA")), (dpath, ("D", "D_B"))):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            torch.save({k: sd[k] for k in keys if k in sd}, path)
        return gpath, dpath

    def load(self, checkpoint_dir):
        """model.py:471-503: load the latest gen/disc checkpoints; False if either is missing."""
        import glob
        gdir, ddir = self._ckpt_paths(checkpoint_dir)
        lg, ld = sorted(glob.glob(gdir + "/cp-*.ckpt")), sorted(glob.glob(ddir + "/cp-*.ckpt"))
        if not (lg and ld):
            return False
        sd = {}
        sd.update(torch.load(lg[-1], map_location="cpu"))
        sd.update(torch.load(ld[-1], map_location="cpu"))
        self.load_state_dict(sd)
        return True

    def test_during_train(self, epoch, args, samples, sink=None):
        """model.py:307-448 (the live part: the CRF variants are commented out there and pydensecrf is out of scope): every
        test sample -- ``samples`` yields (name, sample_image (H,W,3) in [0,1], seg_image (H,W,3) in [0,1]); reading and
        resizing the files (utils.load_test_data) stays on the caller's side -- is rescaled like the reference does
        (tf.image.convert_image_dtype -> uint8 -> float32), translated by the generator, optionally saved under
        ``args.test_dir``, and labelled against its segmentation image (metric.scores_seg_fake); the FCN scores of all labels
        go to ``sink`` under the reference's scalar names.  Returns (the stacked fake images as model.py:440-448 does, scores)."""
        from . import metric as M
        from .utils import convert_image_dtype_uint8, get_img, save_images
        import os
        gts, preds, outputs = [], [], []
        test_dir = getattr(args, "test_dir", None)
        for name, sample_image, seg_image in samples:
            rescaled = convert_image_dtype_uint8(np.asarray(sample_image)[None])                 # :352-353
            fake_A = self.generator(torch.as_tensor(rescaled).to(self.device))                   # :357
            if test_dir:
                os.makedirs(test_dir, exist_ok=True)
                save_images(fake_A, [1, 1], os.path.join(test_dir, os.path.basename(name)))      # :362-365
            fake_img = get_img(fake_A, [1, 1])                                                   # :369
            outputs.append(fake_img)
            lt, lp = M.scores_seg_fake(np.asarray(seg_image, dtype=np.float32)[None], torch.as_tensor(fake_img.astype(np.float32)))   # :373
            preds += list(lp); gts += list(lt)
        score = M.scores(gts, preds, n_class=args.segment_class)                                 # :378
        if sink is not None:                                                                     # :389-393
            sink.scalar("Overall Accuracy", score["Overall Acc"], epoch)
            sink.scalar("Mean Accuracy", score["Mean Acc"], epoch)
            sink.scalar("Frequency Weighted Accuracy", score["FreqW Acc"], epoch)
            sink.scalar("Mean IoU", score["Mean IoU"], epoch)
        return (np.concatenate(outputs, axis=0) if outputs else None), score

    def test(self, args, samples, log=print):
        """model.py:535-567 (--phase test): load the latest checkpoint, translate every test sample and save the input and the
        translation under ``args.test_dir``.  ``samples`` yields (name, sample_image (H,W,3) in [0,1])."""
        from .utils import convert_image_dtype_uint8, save_images
        import os
        log(" [*] Running Test ...")
        log(" [*] Load SUCCESS" if self.load(args.checkpoint_dir) else " [!] Load failed...")
        os.makedirs(args.test_dir, exist_ok=True)
        out = []
        for item in samples:
            name, sample_image = item[0], np.asarray(item[1], dtype=np.float32)
            log("Processing image: " + name)
            rescaled = convert_image_dtype_uint8(sample_image[None])
            fake_A = self.generator(torch.as_tensor(rescaled).to(self.device))
            save_images(sample_image[None], [1, 1], os.path.join(args.test_dir, "real_" + os.path.basename(name)))
            save_images(fake_A, [1, 1], os.path.join(args.test_dir, os.path.basename(name)))
            out.append(fake_A)
        return out

    def train(self, args, batches, log=print, test_samples=None, sink=None):
        """The reference's epoch loop (model.py:202-275) around ``train_step`` for a caller-supplied batch source:
        ``batches(epoch)`` yields dicts with real_A / seg_A / mask_A (+ real_B / seg_B / mask_B in cycle mode) --
        disk loading and augmentation (utils.py:167-233) stay on the caller's side; ``test_samples`` / ``sink`` add the epoch-end
        test pass and the scalar summaries of model.py:263-268.  Prints the reference's line
        (model.py:260), keeps its running-mean loss metrics (model.py:23-24,193-194,270-271), honours
        --continue_train (model.py:210-215) and saves in ``finally`` like model.py:272-275."""
        import time
        start = time.time()
        if getattr(args, "continue_train", False):
            log(" [*] Loading pretrained weights ...")
            log(" [*] Load SUCCESS" if self.load(args.checkpoint_dir) else " [!] Load failed...")
        else:
            log(" [*] New training STARTED")
        history, epoch = [], 0
        try:
            for epoch in range(args.epoch):
                self.gen_loss_metric = self.disc_loss_metric = 0.0
                self._metric_n = 0
                data = list(batches(epoch))
                for idx, b in enumerate(data):
                    for k, v in b.items():
                        setattr(self, k, v)
                    self.train_step(args)
                    gl, dl = self.losses()
                    self._metric_n += 1
                    self.gen_loss_metric += (gl - self.gen_loss_metric) / self._metric_n
                    self.disc_loss_metric += (dl - self.disc_loss_metric) / self._metric_n
                    log("Epoch: [%2d] [%4d/%4d] time: %4.4f Gen_Loss: %f Disc_Loss: %f " % (epoch, idx, len(data), time.time() - start, gl, dl))
                # epoch end (model.py:263-268): the test pass + image summary, then the two running-mean loss scalars
                if test_samples is not None:
                    fake, _ = self.test_during_train(epoch, args, test_samples(epoch) if callable(test_samples) else test_samples, sink)
                    if sink is not None and fake is not None:
                        sink.image("Segmentation Epoch {}".format(epoch), fake, epoch)
                if sink is not None:
                    sink.scalar("Generator Loss", self.gen_loss_metric, epoch)
                    sink.scalar("Discriminator Loss", self.disc_loss_metric, epoch)
                history.append({"epoch": epoch, "Generator Loss": self.gen_loss_metric, "Discriminator Loss": self.disc_loss_metric})
        finally:
            if getattr(args, "checkpoint_dir", None):
                self.save(args.checkpoint_dir, epoch)
        return history


class _PendingReduce:
    """Handle of one network's gradient all-reduces (one per bucket): ``wait()`` orders the compute stream behind all of
    them (a host action, see _host)."""

    def __init__(self, model):
        self._model, self.works = model, []

    def wait(self):
        def wait_all():
            for w in self.works:
                w.wait()
            self.works.clear()
        self._model._host(wait_all)


class _LazyUnpad:
    """``self.fake_A``: the generator output kept in its internal layout; converts to the reference's
    (N,H,W,3) float32 on demand so the step itself does no extra pass."""

    def __init__(self, internal, c):
        self._t, self._c = internal, c

    def tensor(self):
        return K.unpad_channels(self._t, self._c)

    def numpy(self):
        return self.tensor().cpu().numpy()

    @property
    def shape(self):
        return tuple(self._t.shape[:-1]) + (self._c,)

    def __array__(self, dtype=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)
