"""How the CPU baseline (oracle/torch_restatement.py, PyTorch-CPU f32) scales with the thread count on the GPU box's host:
reference-mode step at N=1, 256x256 and 256x512, warm-up + 3 timed steps per thread count.  Picks the pinned count bench.py uses."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sggan_oracle as O            # noqa: E402
from oracle import torch_restatement as T       # noqa: E402


def run(H, W, threads, cycle=False):
    torch.set_num_threads(threads)
    rng = np.random.default_rng(19)
    gs, ds = O.generator_param_shapes(), O.discriminator_param_shapes()
    img = lambda: rng.uniform(0, 1, (1, H, W, 3)).astype(np.float32)
    mh, mw = O.disc_out_hw(H, W)
    mk = lambda: np.stack([O.one_hot(rng.integers(0, 34, (mh, mw)), 34)]).astype(np.float32)
    if cycle:
        P = {n: O.init_params(sh, rng) for n, sh in (("Gab", gs), ("Gba", gs), ("Da", ds), ("Db", ds))}
        S, inputs = T.CycleStep(P, torch.float32), (img(), img(), img(), img(), mk(), mk())
    else:
        S, inputs = T.RefStep(O.init_params(gs, rng), O.init_params(ds, rng), torch.float32), (img(), img(), mk())
    t0 = time.time(); S.step(*inputs); warm = time.time() - t0
    ts = []
    for _ in range(3 if not cycle else 2):
        t1 = time.time(); S.step(*inputs); ts.append(time.time() - t1)
    return warm, ts


if __name__ == "__main__":
    aff = len(os.sched_getaffinity(0))
    print(f"os.cpu_count()={os.cpu_count()} affinity={aff} torch default threads={torch.get_num_threads()}", flush=True)
    for threads in [int(t) for t in (sys.argv[1:] or ["8", "16", "32", "64", "128"])]:
        for H, W in ((256, 256), (256, 512)):
            warm, ts = run(H, W, threads)
            print(f"threads {threads:4d}  reference-mode {W}x{H}: warm {warm:.2f} s, steps {[round(t, 2) for t in ts]} s, median {np.median(ts):.2f}", flush=True)
        warm, ts = run(256, 512, threads, cycle=True)
        print(f"threads {threads:4d}  cycle-mode     512x256: warm {warm:.2f} s, steps {[round(t, 2) for t in ts]} s, median {np.median(ts):.2f}", flush=True)
