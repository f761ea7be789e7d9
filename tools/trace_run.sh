#!/bin/bash
# kernel trace of a short bench run + overlap / per-step duration report (run ON the GPU box): bash tools/trace_run.sh <out-name> [bench args]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/run.log 2>&1
python3 $R/tools/trace_gaps.py $(ls $O/raw/*/*kernel_trace.csv | head -1) > $O/gaps.txt 2>&1
rm -rf $O/raw
