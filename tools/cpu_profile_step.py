"""cProfile of the host side of the train step (where does the enqueue time go).  python tools/cpu_profile_step.py [--mode cycle]"""
import argparse, cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sggan_amd

ap = argparse.ArgumentParser(); ap.add_argument("--mode", default="cycle"); ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
m = sggan_amd.sggan(sggan_amd.default_args(dtype="bf16", device="cuda:0", image_height=256, image_width=512, batch_size=8, cycle=(a.mode == "cycle")))
bench.set_inputs(m, 8, 256, 512, 19)
for _ in range(3):
    m.train_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    m.train_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(25)
