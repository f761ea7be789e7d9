#!/bin/bash
# HBM traffic of the three residual-conv kernels from rocprofv3 PMC passes (run ON the GPU box):
#   bash tools/pmc_traffic.sh        -> gpurun_out/pmc/traffic.json (copy to profiles/ to have bench.py report it)
# One counter set per run, with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for op in fwd dgrad wgrad; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    d=$R/gpurun_out/pmc/${op}_$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/bench_conv.py --ops $op --iters 8 > $d.log 2>&1 || echo "pass $op $c failed"
  done
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc > $R/gpurun_out/pmc/traffic.json
cat $R/gpurun_out/pmc/traffic.json
