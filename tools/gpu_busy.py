"""GPU busy fraction from a rocprofv3 --kernel-trace CSV: sum of kernel durations / wall span, over the densest
contiguous window of the trace (skips start-up).  usage: python tools/gpu_busy.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = len(rows)
print("kernels", n)
# window: middle 50 % of launches
a, b = n // 4, 3 * n // 4
span = rows[b][1] - rows[a][0]
busy = sum(e - s for s, e, _ in rows[a:b + 1])
gaps = [rows[i + 1][0] - rows[i][1] for i in range(a, b)]
gaps_pos = [g for g in gaps if g > 0]
print(f"window {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms = {busy/span:.3f}; launches {b-a+1}; mean gap {sum(gaps_pos)/max(1,len(gaps_pos))/1e3:.2f} us over {len(gaps_pos)} gaps; "
      f"gaps > 10us: {sum(1 for g in gaps if g > 10000)} totalling {sum(g for g in gaps if g > 10000)/1e6:.2f} ms")
big = sorted(((rows[i + 1][0] - rows[i][1], rows[i][2][:60], rows[i + 1][2][:60]) for i in range(a, b)), reverse=True)[:12]
for g, k0, k1 in big:
    print(f"  gap {g/1e3:8.1f} us after {k0} before {k1}")
