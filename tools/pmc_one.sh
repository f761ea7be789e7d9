#!/bin/bash
# PMC counters of one kernel (run ON the GPU box): bash tools/pmc_one.sh <kernel-substring> "<counters>" <script> [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
KN=$1; CTR=$2; shift; shift
O=$R/gpurun_out/pmc_one
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $O/raw -- python3 $R/"$@" > $O/run.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KN" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(f"{k:32s} launches {len(v):3d} mean {sum(v)/len(v):.4g}")
PY
