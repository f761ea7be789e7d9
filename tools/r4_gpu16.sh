#!/bin/bash
# round-4 GPU call 16: phase stamps and ablation timings of the shipped 3x3 halo GEMM (lab build), forward and REFLECT data gradient, 8-image launches
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for op in fwd dgrad; do
  SGG_ABLATE=8 python tools/halo_phases.py $op 2>/dev/null > $O/r4_phases_$op.txt
  cat $O/r4_phases_$op.txt
done
for abl in 0 1 2 3 5 6; do
  echo "== SGG_ABLATE=$abl"
  SGG_LIB_PATH=$R/sg-gan-tf2_amd/libsggan_lab.so SGG_ABLATE=$abl python tools/bench_conv.py --n 8 --iters 60 --ops fwd,dgrad 2>/dev/null
done > $O/r4_ablation.txt 2>&1
cat $O/r4_ablation.txt
