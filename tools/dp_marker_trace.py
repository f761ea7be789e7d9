"""Where the data-parallel step's bucket launches land on the GPU timeline, on ONE GPU (VERDICT r03 #7a).

RCCL elides a single-rank all-reduce (no kernel: a kernel trace of `bench.py` under torch.distributed at world size 1 shows none,
profiles/r04_dp_one_rank.txt), and two RCCL ranks cannot share one device.  So the collective is stood in for by what it is to the
timeline: a stream-ordered operation on a SIDE stream that first waits for the compute stream's position at the launch (as RCCL's
stream does) and then moves the bucket once (a copy of the bucket into a scratch buffer through a multiply kernel -- the name
`MulFunctor` appears nowhere else in the step).  Everything else is the real data-parallel step: RCCL process group at world size 1,
`enable_data_parallel()`, HIP-graph replay with the bucket launches and waits as host actions between graph segments.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/dp_marker_trace.py
    python tools/dp_overlap_trace.py <dir>/*/*kernel_trace.csv MulFunctor
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
import sggan_amd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29551")
fd = os.dup(1); os.dup2(2, 1)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = torch.zeros(1, device="cuda:0"); dist.all_reduce(w); torch.cuda.synchronize()
os.dup2(fd, 1)

m = sggan_amd.sggan(sggan_amd.default_args(dtype="bf16", device="cuda:0", image_height=256, image_width=512, batch_size=8, cycle=True, graph=True))
m.enable_data_parallel()
side = torch.cuda.Stream()
scratch = torch.empty(48 * 2**20 // 4, dtype=torch.float32, device="cuda:0")
orig = m._dp.allreduce_async
sizes = []


def marked(buf):
    work = orig(buf)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.mul(buf, 1.0, out=scratch[:buf.numel()])
    sizes.append(buf.numel() * 4)
    return work


m._dp.allreduce_async = marked
bench.set_inputs(m, 8, 256, 512, 19)
for _ in range(8):
    m.train_step()
torch.cuda.synchronize()
per_step = len(sizes) // 8
print("bucket launches per step:", per_step, " bytes:", [s for s in sizes[-per_step:]])
dist.destroy_process_group()
