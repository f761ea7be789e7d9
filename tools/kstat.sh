#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python tool (run ON the GPU box): bash tools/kstat.sh <out-name> <script> [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kstat_$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/"$@" > $O/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/raw/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int("${KSTAT_ROWS:-80}")]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
PY
rm -rf $O/raw
