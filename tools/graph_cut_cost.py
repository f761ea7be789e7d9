"""Host-side cost of the data-parallel step's host actions ("graph cuts": the bucket launches and waits that run between the
recorded HIP-graph segments, graph.StepProgram.host) -- VERDICT r03 #7(b): the 8-rank step's host budget before the node exists.

One process, RCCL at world size 1 (MASTER_* set here), the bench shape.  Measures, per step: host time to enqueue the replayed
step without data parallelism (one graph segment chain, no cuts) and with it (2 discriminator buckets + (g_buckets + 1) layer
groups per generator pair + 4 waits), the number of segments / host actions, and GPU step time both ways.
    python tools/graph_cut_cost.py [--steps 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
import sggan_amd

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
fd = os.dup(1); os.dup2(2, 1)                      # RCCL's banner goes to stderr
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = torch.zeros(1, device="cuda:0"); dist.all_reduce(w); torch.cuda.synchronize()
os.dup2(fd, 1)


def run(dp):
    m = sggan_amd.sggan(sggan_amd.default_args(dtype="bf16", device="cuda:0", image_height=256, image_width=512, batch_size=8, cycle=True, graph=True))
    if dp:
        m.enable_data_parallel()
    bench.set_inputs(m, 8, 256, 512, 19)
    for _ in range(4):
        m.train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.train_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    prog = m._program
    nseg = sum(1 for kind, _ in prog.items if kind == "graph")
    nhost = sum(1 for kind, _ in prog.items if kind == "host")
    return 1e3 * (t1 - t0) / a.steps, 1e3 * (t2 - t0) / a.steps, nseg, nhost


e0, s0, g0, h0 = run(False)
e1, s1, g1, h1 = run(True)
print(f"graph replay, no data parallelism : host enqueue {e0:.3f} ms/step, step {s0:.3f} ms, graph segments {g0}, host actions {h0}")
print(f"graph replay, RCCL world size 1   : host enqueue {e1:.3f} ms/step, step {s1:.3f} ms, graph segments {g1}, host actions {h1}")
if h1 > 0:
    print(f"host cost per host action (bucket launch or wait, incl. the extra graph launch of its segment): {(e1 - e0) / h1 * 1e3:.1f} us; "
          f"step time with the cuts {100 * (s1 / s0 - 1):+.2f} %")
dist.destroy_process_group()
