#!/bin/bash
# Where the c2-shape forward conv (64 -> 128, stride 2, 256x512 input, generic 256x128 tile) spends its time: HBM traffic, L2 hit
# rate, wave states (run ON the GPU box): bash tools/pmc_c2.sh > gpurun_out/pmc_c2.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="tools/bench_conv.py --n 8 --h 256 --w 512 --c 64 --k 128 --stride 2 --pad SAME --ops fwd --iters 8"
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
  bash $R/tools/pmc_one.sh conv_gemm_glds "$c" $ARGS
done
