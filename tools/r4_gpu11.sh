#!/bin/bash
# round-4 GPU call 11: Conv2DTranspose forward with the norm-statistics epilogue -- op test, full suite, step A/B against the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "deconv" > $O/r4_tests11a.log 2>&1
tail -3 $O/r4_tests11a.log
grep -q "failed\|error" $O/r4_tests11a.log && exit 1
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/r4_all11.log 2>&1
tail -4 $O/r4_all11.log
for i in 1 2 3; do
  for fs in 0 1; do
    echo -n "deconv stats epilogue $fs  "
    SGG_BENCH_DECONV_STATS=$fs timeout -k 10 200 python tools/ab_deconv_stats.py $fs 2>/dev/null | tail -n 1
  done
done > $O/r4_ab_deconv_stats.txt 2>&1
cat $O/r4_ab_deconv_stats.txt
