#!/bin/bash
# Counters of the 7x7 stem / head kernels at the bench shape (run ON the GPU box): bash tools/pmc_7x7.sh > gpurun_out/pmc_7x7.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "stem --c 3 --k 64" "head --c 64 --k 3"; do
  set -- $cfg; name=$1; shift
  ARGS="tools/bench_conv.py --n 8 --h 256 --w 512 --r 7 --pad REFLECT-3 $* --iters 6 --ops fwd,dgrad,wgrad"
  for kn in conv_halo_narrow_in conv7_narrow_out wgrad7_kernel; do
    echo "== $name / $kn"
    for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
      bash $R/tools/pmc_one.sh $kn "$c" $ARGS
    done
  done
done
