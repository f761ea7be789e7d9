"""world_size-2 gloo test of the data-parallel bucket logic (CPU): one all-reduce(sum) per network bucket with
grad_scale = 1/world reproduces the concatenated-batch gradient, and ranks stay bit-identical."""
import os
import socket

import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import sggan_amd  # noqa: F401
    from sggan_amd.dp import GradExchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = GradExchange()
    assert ex.world == world and ex.grad_scale == 1.0 / world
    torch.manual_seed(0)
    n = 1000
    theta = torch.randn(n) + rank                               # replicas differ until rank 0's are broadcast
    ex.broadcast_(theta)
    data = torch.randn(world, 4, n)                             # global batch = world x 4 samples
    loss_grad = lambda x: (2 * (theta - x)).mean(0)             # d/dtheta mean_i |theta - x_i|^2
    g_local = loss_grad(data[rank])
    buckets = {"D": g_local[:400].clone(), "G": g_local[400:].clone()}   # one flat bucket per network
    handles = [ex.allreduce_async(b) for b in buckets.values()]         # the calls sggan.train_step makes
    for h in handles:
        h.wait()
    g = torch.cat([buckets["D"], buckets["G"]]) * ex.grad_scale          # grad_scale is applied inside sgg_adam
    g_ref = loss_grad(data.reshape(-1, n))
    gathered = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(gathered, g)
    if rank == 0:
        out.put((float((g - g_ref).abs().max()), bool(all(torch.equal(gathered[0], t) for t in gathered))))
    dist.destroy_process_group()


def test_dp_allreduce_equals_concatenated_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, same = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert err < 1e-6 and same
