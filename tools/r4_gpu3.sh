#!/bin/bash
# round-4 GPU call: probe, full GPU suite, A/B of the halo GEMM epilogue / patch layout, CPU-baseline thread sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/permlane16_swap.hip -o /tmp/pl16 2>/dev/null && /tmp/pl16 > $O/r4_permlane.txt 2>&1
cat $O/r4_permlane.txt | tail -2
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/r4_all3.log 2>&1
tail -4 $O/r4_all3.log
grep -q " passed" $O/r4_all3.log || exit 1
grep -q "failed" $O/r4_all3.log && exit 1
bash tools/ab_conv.sh "--n 16 --iters 60 --ops fwd_pair,dgrad_pair" libsggan.so libsggan_r3.so libsggan_w1p0.so > $O/r4_ab_wide.txt 2>&1
tail -30 $O/r4_ab_wide.txt
timeout -k 10 500 python tools/cpu_baseline_threads.py 16 32 64 > $O/r4_cpu_threads.txt 2>&1
cat $O/r4_cpu_threads.txt
