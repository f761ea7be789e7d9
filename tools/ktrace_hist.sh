#!/bin/bash
# per-dispatch durations of kernels whose name contains $1 in a short cycle-step run (run ON the GPU box): histogram by grid size
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ktrace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/bench.py --steps 6 --warmup 2 --graph 0 --no-cpu-baseline --no-f32-leg --no-reference-leg > $O/run.log 2>&1
python3 - "$1" <<PY
import csv, glob, sys, collections
pat = sys.argv[1]
f = glob.glob("$O/raw/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
g = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"][:40], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k}: calls {len(v):4d}  avg {sum(v)/len(v):7.1f} us  total {sum(v)/1e3:7.2f} ms ({sum(v)/tot*100:4.1f} %)")
PY
rm -rf $O/raw
