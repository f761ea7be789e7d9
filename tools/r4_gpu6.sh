#!/bin/bash
# round-4 GPU call 6: staggered halves of the halo GEMM (exactness + A/B), static image pool under graph replay, full suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
L=sg-gan-tf2_amd
cp $L/libsggan.so /tmp/main.so
cp $L/libsggan_stg.so $L/libsggan.so
timeout -k 10 600 python -m pytest tests/test_gpu_exact.py tests/test_gpu_ops.py -q -m gpu -x > $O/r4_exact_stg.log 2>&1
cp /tmp/main.so $L/libsggan.so
tail -3 $O/r4_exact_stg.log
bash tools/ab_conv.sh "--n 16 --iters 60 --ops fwd_pair,dgrad_pair" libsggan.so libsggan_stg.so libsggan_stgA.so libsggan_stgL0.so > $O/r4_ab_stagger.txt 2>&1
grep -v "^$" $O/r4_ab_stagger.txt | tail -40
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/r4_all6.log 2>&1
tail -4 $O/r4_all6.log
