#!/usr/bin/env python3
"""bench.py -- train-step images/sec of the SG-GAN hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" (default ``--mode cycle``) is the north_star unit of work: one G_A->B + G_B->A + D_A + D_B
forward/backward/Adam step (4 generator passes, 4 discriminator passes, LSGAN + cycle-L1 + gradient-sensitive
losses; ``sggan._train_step_cycle``) over one synthetic (A,B) batch already resident in HBM; one *image* = one
A sample (a B sample is consumed with it).  ``--mode reference`` times the literal ``model.py:169-200`` step
(1 G + 1 D, paired losses) instead; at N=1 it is also reported beside the headline as ``reference_mode_step``.
Workload = BASELINE.json configs[2]: 512x256 (W x H) images, batch 8 per GPU, bf16 storage / f32 accumulate,
9-block ResNet generators; weak scaling (8 images per GPU), one RCCL all-reduce per network bucket.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying two extra objects:
  roofline     -- the dominant kernel (3x3 C=256 residual-block conv forward, implicit GEMM M=N*64*128, N=256,
                  K=2304): algorithmic FLOPs per launch / its average duration measured with HIP events on the
                  launch stream inside the timed region, vs the 2.5 PFLOP/s dense bf16 MFMA peak.
  cpu_baseline -- the PyTorch-CPU float32 restatement of the same step (oracle/torch_restatement.py; "port":
                  the reference's TF2 path cannot run here) timed on the host cores, N=1 image, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0          # HBM3E spec
# conv MACs per image, forward: G(H,W) scales with H*W; D from its layer table
G_GMAC_256x512 = 99.103
D_GMAC = {(128, 128): 0.599, (256, 256): 2.550, (256, 512): 5.189, (512, 1024): 21.375}


def step_gflop_per_image(H, W, mode="reference"):
    g = G_GMAC_256x512 * (H * W) / (256 * 512)
    d = D_GMAC.get((H, W))
    if d is None:
        d = 5.189 * (H * W) / (256 * 512)
    if mode == "cycle":
        # 4 G fwd + 4 G bwd (2x) = 12G;  D: 4 fwd + 4 full bwd (2x) + 2 data-gradient passes for the G loss = 14D
        # (SURVEY.md 8(d) counts 16D: it evaluates D on the fakes twice; the build evaluates them once -- D2)
        return 2.0 * (12 * g + 14 * d)
    return 2.0 * (3 * g + 7 * d)          # reference-mode step: 3G + 7D MACs


def synthetic_batch(model, N, H, W, seed):
    """real ~ U[0,1); class-index map = 32x32-px blocks over 16 label ids; seg = palette
    colour / 255; mask = one-hot of the index map resampled to D's output grid (deviation D1)."""
    import sggan_amd.segment_class as sc
    g = torch.Generator().manual_seed(seed)
    real = torch.rand((N, H, W, 3), generator=g)
    ids = torch.tensor([0, 1, 3, 4, 7, 8, 11, 17, 20, 21, 22, 23, 24, 25, 26, 33])
    blocks = ids[torch.randint(0, len(ids), (N, H // 32, W // 32), generator=g)]
    idx = blocks.repeat_interleave(32, 1).repeat_interleave(32, 2).to(torch.uint8)
    palette = torch.randint(0, 256, (34, 3), generator=torch.Generator().manual_seed(1234)).float() / 255.0
    seg = palette[idx.long()]
    mh, mw = model.discriminator.out_hw(H, W)
    if (mh, mw) == (1, 1):
        mh, mw = round(H / 34), round(W / 34)        # the reference's loader grid (utils.py:197-199)
    mask = sc.one_hot_mask(idx.to(model.device), mh, mw, 34)
    real_i = model.generator.to_internal(real.to(model.device))
    seg_i = model.generator.to_internal(seg.to(model.device))
    return real_i, seg_i, mask, (real, seg, mask.cpu())


def set_inputs(model, N, H, W, seed):
    a = synthetic_batch(model, N, H, W, seed)
    model.real_A, model.seg_A, model.mask_A = a[0], a[1], a[2]
    if model.cycle:
        b = synthetic_batch(model, N, H, W, seed + 1000)
        model.real_B, model.seg_B, model.mask_B = b[0], b[1], b[2]


TRAFFIC_FILES = ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json")        # newest first


def measured_traffic(kernel, gflop_per_launch=None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json: FETCH_SIZE doubled per
    MI355X_MICROARCH.md + WRITE_SIZE, separate --pmc passes on tools/bench_conv.py).  Returns (bytes, note); an entry collected
    on a launch of another size (its "gflop_per_launch") is scaled to this one and the note says so; (None, None) if not collected."""
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                table = json.load(f)
        except Exception:
            continue
        # the entry collected on the launch size closest to this one ("..._pair": the 16-image launches of the paired cycle step)
        cands = [k for k in (kernel, kernel + "_pair") if k in table]
        if not cands:
            continue
        key = min(cands, key=lambda k: abs(table[k].get("gflop_per_launch", 77.309411328) - (gflop_per_launch or 77.309411328)))
        e = table[key]
        b = e["hbm_bytes_per_launch"]                       # (helper kernels of the call are listed separately in the file)
        g = e.get("gflop_per_launch", 77.309411328)
        note = f"FETCH_SIZE (doubled, gfx950) + WRITE_SIZE per launch of the main kernel, profiles/{name} [{key}]"
        if gflop_per_launch is not None and abs(g - gflop_per_launch) > 1e-3 * g:
            b = b * gflop_per_launch / g
            note += f" (collected on a {g:.1f} GFLOP launch, scaled to this launch's {gflop_per_launch:.1f})"
        return b, note
    return None, None


def measured_in_traffic(algorithmic_bytes):
    """HBM bytes of the instance-norm apply pass per algorithmic byte (profiles/r02_in_traffic.json: 33.55 MB tensor read + written),
    scaled to `algorithmic_bytes`; None if not collected."""
    for name in ("r03_in_traffic.json", "r02_in_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)["in_apply_kernel_fwd"]["hbm_bytes_per_launch"] / (2 * 33554432.0) * algorithmic_bytes
        except Exception:
            continue
    return None


class EventProfiler:
    """Times selected launches with HIP events on the stream the kernels are launched on, inside the timed region.

    The events ride on the kernel's OWN dispatch packet (libsggan's sgg_time_next_launch -> hipExtLaunchKernel start / stop
    events), so a span is that kernel's begin-to-end time as the command processor stamps it -- the clock rocprofv3's kernel
    trace reads -- with no marker packets in the queue and nothing to subtract (round 1 bracketed the call with torch.cuda.Event
    markers and took an empty pair's cost off; both read the same within 1 %).  For a call that launches helpers too (side-tensor
    gather, slab reduce, norm finalize) the span is the MAIN kernel only.  NOTE for readers of the rocprofv3 summary of this
    command: the reference-mode leg at the end of the run (`reference_mode_step`) launches the same kernels on 8 images where the
    cycle step's paired launches cover 16, so a per-name average mixes the two unless the name says which (the paired
    instantiations carry PAIR = true as their last template argument).
    Events cannot be read back from inside a captured HIP graph, so in graph mode the LAST steps of the timed region are
    dispatched eagerly -- same kernels, same order, same state, same throughput -- and those are the steps whose launches are
    timed.  ``mode``: "off" or "live"."""

    def __init__(self, res_hw=(64, 128)):
        self.records = {}
        self.mode = "off"
        self.res_hw = tuple(res_hw)          # spatial size of the residual blocks (H/4, W/4)
        self.unused = 0                      # armed pairs no timed-family launch picked up (a call that took another kernel path)

    class _Span:
        def __init__(self, prof, store):
            import ctypes
            from sggan_amd import _abi as A
            self.prof, self.store, self.A = prof, store, A
            self.s, self.e = ctypes.c_void_p(), ctypes.c_void_p()
            A.check(A.lib().sgg_event_create(ctypes.byref(self.s)), "event_create")
            A.check(A.lib().sgg_event_create(ctypes.byref(self.e)), "event_create")

        def start(self):
            self.A.lib().sgg_time_next_launch(self.s, self.e)

        def stop(self):
            if self.A.lib().sgg_time_next_launch(None, None) == 1:
                self.store.append((self.s, self.e))
            else:
                self.prof.unused += 1
                self.A.lib().sgg_event_destroy(self.s); self.A.lib().sgg_event_destroy(self.e)

    def __call__(self, name, key):
        if self.mode == "off":
            return None
        tag = None
        if name in ("conv2d_fwd", "conv2d_fwd_pair") and hasattr(key, "desc"):      # _pair: both generators' images in one launch
            d = key.desc
            if d.R == 3 and d.C == 256 and d.K == 256 and d.stride == 1 and d.pad_mode == 1:
                tag = ("res_conv_fwd", d.N * d.Ho * d.Wo, d.K, d.R * d.S * d.C)
        elif name in ("conv2d_bwd_data", "conv2d_bwd_data_pair") and hasattr(key, "desc"):
            d = key.desc
            if d.R == 3 and d.C == 256 and d.K == 256 and d.stride == 1 and d.pad_mode == 1:
                tag = ("res_conv_dgrad", d.N * d.H * d.W, d.C, d.R * d.S * d.K)
        elif name == "conv2d_bwd_weight" and hasattr(key, "desc"):
            d = key.desc
            if d.R == 3 and d.C == 256 and d.K == 256 and d.stride == 1 and d.pad_mode == 1:
                tag = ("res_conv_wgrad", d.R * d.S * d.C, d.K, d.N * d.Ho * d.Wo)
        elif name == "conv2d_bwd_weight_pair" and hasattr(key, "desc"):
            d = key.desc
            if d.R == 3 and d.C == 256 and d.K == 256 and d.stride == 1 and d.pad_mode == 1:
                tag = ("res_conv_wgrad_pair", d.R * d.S * d.C, d.K, 2 * d.N * d.Ho * d.Wo)   # two applications, one launch
        elif name == "conv2d_bwd_weight_pair2" and hasattr(key, "desc"):
            d = key.desc
            if d.R == 3 and d.C == 256 and d.K == 256 and d.stride == 1 and d.pad_mode == 1:
                tag = ("res_conv_wgrad_pair2", d.R * d.S * d.C, d.K, 4 * d.N * d.Ho * d.Wo)   # two networks x two applications, one launch
        elif name in ("instnorm_fwd", "instnorm_fwd_pair", "instnorm_fwd_partial", "instnorm_fwd_partial_pair"):
            shape, has_res = key
            if len(shape) == 4 and shape[3] == 256 and tuple(shape[1:3]) == self.res_hw:
                # the apply pass: read x (+ the skip tensor in the block's second norm), write y
                tag = ("res_instnorm_apply_fwd", shape[0] * shape[1] * shape[2] * shape[3], 3 if has_res else 2)
        if tag is None:
            return None
        return self._Span(self, self.records.setdefault(tag, []))

    def summary(self):
        """tag -> (mean kernel time in ms, samples); call after a device synchronize"""
        import ctypes
        from sggan_amd import _abi as A
        out = {}
        for tag, evs in self.records.items():
            ms = []
            for s, e in evs:
                v = ctypes.c_float()
                A.check(A.lib().sgg_event_elapsed_ms(s, e, ctypes.byref(v)), "event_elapsed")
                ms.append(v.value)
                A.lib().sgg_event_destroy(s); A.lib().sgg_event_destroy(e)
            if ms:
                out[tag] = (float(np.mean(ms)), len(ms))
        self.records = {}
        return out


CPU_THREADS = 16        # the CPU leg's pinned thread count (see cpu_baseline): a one-GPU box's CPU share


def cpu_baseline(H, W, seed, mode):
    """The PyTorch-CPU f32 restatement (oracle/torch_restatement.py; kind "port" -- TensorFlow is not installed, so the reference's
    own TF2-CPU path cannot be timed) on the host cores, N = 1, PINNED to CPU_THREADS threads: with PyTorch's default (128 threads
    on this 256-CPU host) oneDNN's N = 1 convolutions oversubscribe and one cycle step took 18-32 s, moving 50 % between runs
    (rounds 1-3); at 16 threads it takes ~2.4 s, at 32 ~3.5 s, at 64 ~6.6 s (tools/cpu_baseline_threads.py,
    gpurun_out/r4_cpu_threads.txt).  Three samples, each the MEDIAN of its timed steps after one warm-up: `value` = the step the
    headline times (cycle mode, or reference mode with --mode reference) at the headline's image size; beside it what SURVEY
    8(d) specified -- the reference-mode step (model.py:169-200) at BASELINE configs[0]'s 256x256 and at 256x512."""
    from oracle import torch_restatement as T
    from oracle import sggan_oracle as O
    before = torch.get_num_threads()
    threads = max(1, min(CPU_THREADS, len(os.sched_getaffinity(0))))
    torch.set_num_threads(threads)
    try:
        gs, ds = O.generator_param_shapes(), O.discriminator_param_shapes()

        def sample(h, w, cycle, n_timed):
            rng = np.random.default_rng(seed)
            img = lambda: rng.uniform(0, 1, (1, h, w, 3)).astype(np.float32)
            mh, mw = O.disc_out_hw(h, w)
            mk = lambda: np.stack([O.one_hot(rng.integers(0, 34, (mh, mw)), 34)]).astype(np.float32)
            if cycle:
                P = {n: O.init_params(sh, rng) for n, sh in (("Gab", gs), ("Gba", gs), ("Da", ds), ("Db", ds))}
                S, inputs = T.CycleStep(P, torch.float32), (img(), img(), img(), img(), mk(), mk())
            else:
                S, inputs = T.RefStep(O.init_params(gs, rng), O.init_params(ds, rng), torch.float32), (img(), img(), mk())
            t0 = time.time()
            S.step(*inputs)
            warm = time.time() - t0
            times = []
            for _ in range(n_timed):
                t1 = time.time()
                S.step(*inputs)
                times.append(time.time() - t1)
            dt = float(np.median(times))
            return {"images_per_sec": 1.0 / dt, "median_step_s": dt, "step_times_s": [round(t, 3) for t in times], "warmup_s": round(warm, 3),
                    "gflops": step_gflop_per_image(h, w, "cycle" if cycle else "reference") / dt,
                    "what": f"{'cycle-mode step (2G+2D)' if cycle else 'reference-mode step (model.py:169-200, 1G+1D)'}, N=1 {w}x{h} f32"}

        head = sample(H, W, mode == "cycle", 3 if mode == "cycle" else 7)
        ref_small = sample(256, 256, False, 9)      # (0.2-0.3 s steps: more of them; between RUNS this sample still moved 0.21 -> 0.28 s)
        ref_full = sample(H, W, False, 7) if mode == "cycle" else head
    finally:
        torch.set_num_threads(before)
    return {"value": head["images_per_sec"], "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"median of {len(head['step_times_s'])} timed steps after 1 warm-up: {head['what']}; PyTorch-CPU restatement "
                      f"(oracle/torch_restatement.py), not TF2; torch threads pinned to {threads} (os.cpu_count()={os.cpu_count()}, "
                      f"affinity {len(os.sched_getaffinity(0))}); step times {head['step_times_s']} s",
            "gflops": head["gflops"],
            "reference_mode_256x256": ref_small, f"reference_mode_{W}x{H}": ref_full}


class _StdoutToStderr:
    """RCCL prints a version banner on stdout when its communicator is created; the bench contract is ONE JSON line
    on stdout, so file descriptor 1 is pointed at stderr while the process group / first collective come up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--mode", default="cycle", choices=["cycle", "reference"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--mixed", type=int, default=0, help="1: bf16 storage with the f32 activation-gradient chain through the residual blocks (sggan mixed=True)")
    ap.add_argument("--no-reference-leg", action="store_true", help="skip the reference-mode step timed beside the cycle headline (profiling: keeps the kernel statistics to the cycle step)")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step from captured HIP graphs (default); 0: eager per-launch dispatch from Python")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the short f32 (the reference's precision) run of the same step reported as f32_step")
    ap.add_argument("--group2", type=int, default=1, help="0: the lockstep pairs launch their generic convolutions once per network instead of as one grouped launch (A/B)")
    ap.add_argument("--d-quad", type=int, default=1, help="0: reals and fakes go through the discriminators as two stacked passes instead of one (A/B)")
    ap.add_argument("--checkpoint-blocks", type=int, default=0, help="1: activation checkpointing of the generators' residual blocks (BASELINE configs[4]): each block is re-run in backward")
    ap.add_argument("--lib", default=None, help="A/B timing: bind this build of the library instead of the in-tree libsggan.so")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with nproc-per-node {a.gpus} (WORLD_SIZE={world})")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: exercise the RCCL path even at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _StdoutToStderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            warm = torch.zeros(1, device=f"cuda:{local}")
            dist.all_reduce(warm)                     # creates the communicator (and prints the banner) now
            torch.cuda.synchronize()

    import sggan_amd
    if a.lib:
        sggan_amd._abi.use_library(a.lib)
    from sggan_amd import kernels as K
    def make_model(mode, dtype=None):
        m = sggan_amd.sggan(sggan_amd.default_args(dtype=dtype or a.dtype, device=f"cuda:{local}", image_height=a.height,
                                                   image_width=a.width, batch_size=a.batch, cycle=(mode == "cycle"), graph=bool(a.graph), mixed=bool(a.mixed), group2=bool(a.group2), d_quad=bool(a.d_quad),
                                                   checkpoint_blocks=bool(a.checkpoint_blocks)))
        if dist is not None:
            m.enable_data_parallel()
        set_inputs(m, a.batch, a.height, a.width, 19 + rank)
        return m

    model = make_model(a.mode)

    prof = EventProfiler((a.height // 4, a.width // 4))
    K.PROFILE = prof
    timing = (rank == 0) and not a.no_kernel_timing
    # graph mode: the last `eager_tail` steps of the timed region are dispatched eagerly so that their launches can be timed
    # with events (EventProfiler); every rank does the same (the steps all-reduce)
    eager_tail = 0 if (a.no_kernel_timing or not a.graph) else max(2, a.steps // 5)
    eager_tail = min(eager_tail, a.steps)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        model.use_graph = bool(a.graph) and i < max(1, a.warmup - 1)      # one eager warm-up step too, when there is room
        model.train_step()
    model.use_graph = bool(a.graph)
    if a.graph and model._program is None:         # recording is not part of the timed region
        model._stage_inputs(); model._record()
    barrier()
    t0 = time.perf_counter()
    t_switch = None
    for i in range(a.steps):
        if a.graph and eager_tail and i == a.steps - eager_tail:
            torch.cuda.synchronize()               # (one queue drain inside the timed region: the replayed and the eagerly
            t_switch = time.perf_counter()         #  dispatched steps are also reported separately)
            model.use_graph = False
        if timing and (not a.graph or i >= a.steps - eager_tail):
            prof.mode = "live"
        model.train_step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof.mode = "off"
    model.use_graph = bool(a.graph)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gl, dl = model.losses()
    if not (np.isfinite(gl) and np.isfinite(dl)):
        raise SystemExit(f"non-finite losses after the timed region: gen {gl} disc {dl}")

    if rank == 0:
        images = a.batch * world * a.steps
        ips = images / elapsed
        gflop_img = step_gflop_per_image(a.height, a.width, a.mode)
        what = ("cycle-mode train step (north_star unit: G_A->B + G_B->A + D_A + D_B, LSGAN + cycle-L1 + gradient-sensitive "
                "losses; 12G+14D conv MACs)" if a.mode == "cycle" else
                "reference-mode train_step (model.py:169-200; 1 G + 1 D, paired losses; 3G+7D conv MACs)")
        line = {
            "metric": "train-step images/sec (G+D fwd/bwd) at 512x256", "value": ips, "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: {what}, 9-block ResNet generators, "
                                   f"{a.width}x{a.height}, batch {a.batch}/GPU, {a.dtype} storage / f32 accumulate", "mode": a.mode,
                       "global_batch": a.batch * world, "height": a.height, "width": a.width, "parallelism": f"dp{world}",
                       "gflop_per_image": gflop_img, "mixed_gradient_chain": bool(a.mixed),
                       "dispatch": (f"HIP-graph replay of the recorded step (sggan_amd/graph.py) for {a.steps - eager_tail} of the {a.steps} timed "
                                    f"steps, eager launches for the last {eager_tail} (the ones whose kernels are timed with events)"
                                    if a.graph else "eager launches from Python")},
            "graph_steps_images_per_sec": (a.batch * world * (a.steps - eager_tail) / (t_switch - t0)) if t_switch else None,
            "eager_tail_images_per_sec": (a.batch * world * eager_tail / (elapsed - (t_switch - t0))) if t_switch else None,
            "step_tflops": ips * gflop_img / 1e3,
            "step_frac_of_mfma_peak": ips * gflop_img / 1e3 / (PEAK_BF16_TFLOPS * world),
            "gen_loss": gl, "disc_loss": dl,
            "peak_device_memory_mib": torch.cuda.max_memory_allocated() / 2**20,
            "activation_checkpointing": bool(a.checkpoint_blocks),
        }
        summ = prof.summary()
        kt = {}
        esz = 2 if a.dtype == "bf16" else 4
        for tag, (ms, n) in summ.items():
            if tag[0].startswith("res_conv"):
                fl = 2.0 * tag[1] * tag[2] * tag[3]
                kt[tag[0]] = {"avg_ms": ms, "launches": n, "tflops": fl / (ms * 1e-3) / 1e12, "gflop_per_launch": fl / 1e9}
            else:
                by = tag[2] * tag[1] * esz                               # (1 or 2) reads + 1 write of the tensor
                k = kt.setdefault(tag[0], {"ms_total": 0.0, "launches": 0, "bytes_total": 0.0})
                k["ms_total"] += ms * n; k["launches"] += n; k["bytes_total"] += by * n
        for k in kt.values():
            if "bytes_total" in k:
                k["avg_ms"] = k["ms_total"] / k["launches"]
                k["gbs"] = k["bytes_total"] / (k["ms_total"] * 1e-3) / 1e9
                k["mbytes_per_launch"] = k.pop("bytes_total") / k["launches"] / 1e6
                k.pop("ms_total")
        how = ("HIP events on the kernel's own dispatch packet (sgg_time_next_launch -> hipExtLaunchKernel start/stop events) on the "
               "launch stream, every launch of this kernel inside the timed region"
               + (f" (its last {eager_tail} steps, which are dispatched eagerly: events cannot be read back from inside a replayed HIP graph)" if a.graph else "")
               + "; helper launches of the same call (side-tensor gather, slab reduce, norm finalize) are not in the span")
        # the three residual-block GEMM families; `roofline` is the one with the largest share of the timed steps
        fam = {"res_conv_fwd": "conv3x3_halo_gemm_kernel<FWD, STATS, PAIR> -- 3x3 C=256 residual-block conv forward (bf16, 256x256 tile, 8 waves, "
                               "input halo resident in LDS); also computes the following instance norm's per-channel sums in its epilogue (~4 us, not in the FLOPs)",
               "res_conv_dgrad": "conv3x3_halo_gemm_kernel<DGRAD, REFLECT, PAIR> -- data gradient of the same conv incl. MirrorPadGrad (virtual rows + column "
                                 "patches from fold_halo_gather_kernel, a ~7 us helper launch not in the span) and the skip-gradient addend in its epilogue",
               "res_conv_wgrad_pair2": "conv3x3_wgrad_halo_kernel -- weight gradient of the same conv, all nine taps per block; in the cycle step ONE launch covers both "
                                       "generators x both applications (sgg_conv2d_bwd_weight_pair2); the f32 slab reduce is a helper launch not in the span",
               "res_conv_wgrad_pair": "conv3x3_wgrad_halo_kernel (two applications of one generator per launch)",
               "res_conv_wgrad": "conv3x3_wgrad_halo_kernel (one application per launch)"}
        roofs = {}
        for name in fam:
            if name not in kt:
                continue
            k = kt[name]
            tr, tnote = measured_traffic(name if name != "res_conv_wgrad_pair" else "res_conv_wgrad", k["gflop_per_launch"])
            roofs[name] = {"bound": "mfma", "kernel": fam[name] + "; in the cycle step one launch covers the stacked images of both generators -- "
                                                                   "gflop_per_launch says how many",
                           "achieved": k["tflops"], "peak": PEAK_BF16_TFLOPS if a.dtype == "bf16" else 157.3, "unit": "TFLOP/s",
                           "frac": k["tflops"] / (PEAK_BF16_TFLOPS if a.dtype == "bf16" else 157.3), "traffic": tr, "traffic_note": tnote,
                           "avg_launch_ms": k["avg_ms"], "launches_timed": k["launches"], "gflop_per_launch": k["gflop_per_launch"],
                           "share_of_timed_steps_ms": k["avg_ms"] * k["launches"], "timing": how}
        if roofs:
            top = max(roofs, key=lambda n: roofs[n]["share_of_timed_steps_ms"])
            line["roofline"] = dict(roofs[top], name=top, why="largest total time among the step's kernels (residual-block GEMM families) in the timed launches")
            line["roofline_others"] = {n: r for n, r in roofs.items() if n != top}
        else:
            line["roofline"] = None
        if "res_instnorm_apply_fwd" in kt:
            k = kt["res_instnorm_apply_fwd"]
            line["roofline_instnorm"] = {"bound": "hbm", "achieved": k["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                         "frac": k["gbs"] / PEAK_HBM_GBS, "traffic": measured_in_traffic(k["mbytes_per_launch"] * 1e6),
                                         "avg_launch_ms": k["avg_ms"], "launches_timed": k["launches"], "mbytes_per_launch": k["mbytes_per_launch"],
                                         "note": "in_apply_kernel forward on the residual blocks' (N,H/4,W/4,256) tensors: normalise + ReLU (read, write) and "
                                                 "normalise + skip add (2 reads, write), averaged over both; the statistics come from the conv epilogue "
                                                 "(finalize launch not in the span); traffic = FETCH_SIZE (doubled) + WRITE_SIZE of the kernel's read+write "
                                                 "form per tensor byte (profiles/r02_in_traffic.json), scaled to the timed mix"}
        line["kernels"] = kt
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.height, a.width, 19, a.mode)
        else:
            line["cpu_baseline"] = None
        if world == 1 and a.dtype != "f32" and not a.no_f32_leg:
            # the same step at the reference's own precision (Keras float32 end to end; module.py layers are default-dtype):
            # f32 storage, v_mfma_f32_16x16x4_f32 (1/16 of the bf16 matrix rate), same kernels otherwise; short run
            K.PROFILE = None
            model._program = None
            del model
            torch.cuda.empty_cache()
            m32 = make_model(a.mode, "f32")
            for _ in range(2):
                m32.train_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n32 = 4
            for _ in range(n32):
                m32.train_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n32
            gl32, dl32 = m32.losses()
            line["f32_step"] = {"images_per_sec": a.batch / dt, "ms_per_step": 1e3 * dt, "steps": n32, "dtype": "f32",
                                "step_tflops": a.batch / dt * gflop_img / 1e3, "step_frac_of_f32_mfma_peak": a.batch / dt * gflop_img / 1e3 / 157.3,
                                "gen_loss": gl32, "disc_loss": dl32,
                                "what": f"the same {a.mode}-mode step, same shapes, f32 storage and f32 MFMA (the reference trains in float32): "
                                        "the same-precision-as-the-reference number beside the bf16 headline"}
            model = m32
        if world == 1 and a.mode == "cycle" and not a.no_reference_leg:
            # the literal reference step (1 G + 1 D) beside the headline, same shapes, short run
            K.PROFILE = None
            model._program = None
            del model
            torch.cuda.empty_cache()
            ref = make_model("reference")
            for _ in range(3):
                ref.train_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nref = max(5, a.steps // 2)
            for _ in range(nref):
                ref.train_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / nref
            gf = step_gflop_per_image(a.height, a.width, "reference")
            line["reference_mode_step"] = {"images_per_sec": a.batch / dt, "ms_per_step": 1e3 * dt, "gflop_per_image": gf,
                                           "step_tflops": a.batch / dt * gf / 1e3, "steps": nref,
                                           "what": "model.py:169-200 (1 G + 1 D, BCE + 100*L1), same shapes and dtype"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
