#!/bin/bash
# round-4 GPU call 8: bucket-launch positions on one GPU (stand-in kernels), wave-state counters of the shipped paired halo GEMMs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/dpmark -- python3 $R/tools/dp_marker_trace.py > $O/dp_marker_run.txt 2> $O/dp_marker_run.err
tail -2 $O/dp_marker_run.txt
python3 $R/tools/dp_overlap_trace.py $(ls $O/dpmark/*/*kernel_trace.csv | head -1) MulFunctor > $O/dp_marker_trace.txt 2>&1
rm -rf $O/dpmark
tail -40 $O/dp_marker_trace.txt
bash $R/tools/pmc_waves.sh dgrad_pair 16 > $O/waves_dgrad_pair.txt 2>&1
bash $R/tools/pmc_waves.sh fwd_pair 16 > $O/waves_fwd_pair.txt 2>&1
cat $O/waves_dgrad_pair.txt | tail -36
