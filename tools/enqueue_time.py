"""CPU enqueue time vs GPU time of the train step: if the host finishes enqueuing a step well before the GPU finishes
running it, launch overhead is hidden and only GPU-side dispatch gaps remain.
    python tools/enqueue_time.py [--mode cycle|reference] [--graph 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sggan_amd

ap = argparse.ArgumentParser(); ap.add_argument("--mode", default="cycle"); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--graph", type=int, default=0)
a = ap.parse_args()
m = sggan_amd.sggan(sggan_amd.default_args(dtype="bf16", device="cuda:0", image_height=256, image_width=512, batch_size=8, cycle=(a.mode == "cycle"), graph=bool(a.graph)))
bench.set_inputs(m, 8, 256, 512, 19)
for _ in range(3):
    m.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    m.train_step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"mode {a.mode} graph={a.graph}: enqueue {1e3*(t1-t0)/a.steps:.2f} ms/step, total {1e3*(t2-t0)/a.steps:.2f} ms/step")
