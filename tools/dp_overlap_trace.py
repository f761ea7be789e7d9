"""Where the data-parallel step's all-reduce kernels land relative to the backward kernels, from a rocprofv3 --kernel-trace CSV of

    RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 rocprofv3 --kernel-trace --output-format csv -d <dir> -- \
        python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-reference-leg --no-f32-leg

(one rank: RCCL moves nothing, but every bucket's all-reduce is a real kernel on RCCL's stream, launched by the same host actions
between the same graph segments as on 8 ranks).  For each step of the timed region: every RCCL kernel's start offset from the
step's first kernel, the compute kernel running at that moment, and how much compute was still queued behind it -- i.e. whether
the bucket launches sit where DESIGN.md section 5 says (discriminator buckets under the generators' backward; generator layer
groups inside the second backward pass; only the stem bucket at the very end).
    python tools/dp_overlap_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [k for k in rows[0] if "Start" in k][0]; ke = [k for k in rows[0] if "End" in k][0]; kn = [k for k in rows[0] if "Kernel_Name" in k][0]
ev = sorted(((int(r[ks]), int(r[ke]), r[kn]) for r in rows), key=lambda t: t[0])
MARK = sys.argv[2] if len(sys.argv) > 2 else None          # stand-in kernels of a one-GPU run (tools/dp_marker_trace.py)
is_cc = lambda n: "nccl" in n.lower() or "rccl" in n.lower() or (MARK is not None and MARK in n)
cc = [e for e in ev if is_cc(e[2])]
comp = [e for e in ev if not is_cc(e[2])]
print(f"{len(ev)} kernels, {len(cc)} of them RCCL" + (f" / {MARK} stand-ins" if MARK else ""))
if not cc:
    sys.exit(0)
# steps: an Adam launch of the last network ends a step; take the LAST complete steps (graph replays) of the trace
adam = [i for i, e in enumerate(comp) if "adam_iter_kernel" in e[2]]
steps = []
for a, b in zip(adam[3::4], adam[7::4]):           # 4 networks per step: spans between every 4th Adam launch
    steps.append((comp[a][1], comp[b][1]))
short = lambda n: n.split("(")[0].replace("void ", "")[:70]
for si, (t0, t1) in enumerate(steps[-3:]):
    inside = [e for e in comp if t0 <= e[0] < t1]
    ccs = [e for e in cc if t0 <= e[0] < t1]
    if not inside:
        continue
    first, last = inside[0][0], inside[-1][1]
    print(f"\nstep {si}: {len(inside)} compute kernels over {(last - first) / 1e6:.2f} ms, {len(ccs)} all-reduce kernels")
    for c in ccs:
        running = [e for e in inside if e[0] <= c[0] < e[1]]
        behind = sum(e[1] - e[0] for e in inside if e[0] >= c[0]) / 1e6
        print(f"  +{(c[0] - first) / 1e6:7.3f} ms  dur {(c[1] - c[0]) / 1e3:7.1f} us  {short(c[2])[:40]:40s} | beside: "
              f"{short(running[0][2]) if running else '(gap)':60s} | compute still to run: {behind:6.2f} ms")
