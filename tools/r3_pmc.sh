#!/bin/bash
# Round-3 PMC set for the launches the paired cycle step actually makes (run ON the GPU box): bash tools/r3_pmc.sh
#   gpurun_out/r3pmc/traffic.json          HBM bytes per launch: fwd_pair / dgrad_pair (16 stacked images), wgrad_pair2 (2 networks x 2 x 8 images)
#   gpurun_out/r3pmc/waves_dgrad_pair.txt  wave-state / instruction-mix counters of the paired REFLECT data gradient
# One counter set per run, --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3pmc
mkdir -p $O/raw
cd /tmp && export TMPDIR=/tmp
for op in fwd_pair dgrad_pair wgrad_pair2; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    d=$O/raw/${op}_$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/bench_conv.py --n 16 --ops $op --iters 8 > $d.log 2>&1 || echo "pass $op $c failed"
  done
done
python3 $R/tools/pmc_traffic.py $O/raw fwd_pair,dgrad_pair,wgrad_pair2 16 > $O/traffic.json
rm -rf $O/raw
cat $O/traffic.json
bash $R/tools/pmc_waves.sh dgrad_pair 16 > $O/waves_dgrad_pair.txt 2>&1
bash $R/tools/pmc_waves.sh fwd_pair 16 > $O/waves_fwd_pair.txt 2>&1
