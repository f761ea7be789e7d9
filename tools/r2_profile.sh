#!/bin/bash
# Round-2 measurement set (run ON the GPU box, one call): bash tools/r2_profile.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2final
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/bench_default_kernel_stats.csv
rm -rf $O/stats
ls -la $O
