"""Per-kernel durations and inter-kernel gaps from a rocprofv3 --kernel-trace CSV: do consecutive kernels of the stream overlap?
   python tools/trace_gaps.py <kernel_trace.csv> [name-fragment ...]"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
frags = sys.argv[2:] or ["conv3x3_halo_gemm_kernel<0, false, 1>", "conv3x3_halo_gemm_kernel<1, true, 0>", "conv3x3_wgrad_halo_kernel", "in_apply_kernel"]
ks = [k for k in rows[0] if "Start" in k][0]; ke = [k for k in rows[0] if "End" in k][0]; kn = [k for k in rows[0] if "Kernel_Name" in k][0]
ev = sorted(((int(r[ks]), int(r[ke]), r[kn]) for r in rows), key=lambda t: t[0])
st = np.array([e[0] for e in ev], dtype=np.int64); en = np.array([e[1] for e in ev], dtype=np.int64)
gap_prev = np.concatenate([[0], st[1:] - en[:-1]])           # start - previous end (negative: overlap with the previous kernel)
gap_next = np.concatenate([st[1:] - en[:-1], [0]])
print(f"{len(ev)} kernels; gap to previous kernel: median {np.median(gap_prev)/1e3:.2f} us, p10 {np.percentile(gap_prev,10)/1e3:.2f}, p90 {np.percentile(gap_prev,90)/1e3:.2f}; "
      f"negative (overlap) in {np.mean(gap_prev < 0)*100:.1f} % of launches")
t0 = st[0]
for f in frags:
    idx = [i for i, e in enumerate(ev) if f in e[2]]
    if not idx:
        continue
    d = (en[idx] - st[idx]) / 1e3
    print(f"\n{f}: {len(idx)} launches, duration mean {d.mean():.1f} us")
    # time-ordered buckets of 36 launches (~ one step's worth for the residual convs)
    nb = max(1, len(idx) // 36)
    for b in range(nb):
        sl = idx[b * 36:(b + 1) * 36]
        dd = (en[sl] - st[sl]) / 1e3
        print(f"  launches {b*36:4d}-{b*36+len(sl)-1:4d}  t={(st[sl[0]]-t0)/1e6:8.1f} ms  dur mean {dd.mean():7.1f} min {dd.min():7.1f} max {dd.max():7.1f} | gap before {gap_prev[sl].mean()/1e3:6.2f} us  after {gap_next[sl].mean()/1e3:6.2f} us")
