#!/bin/bash
# round-4 GPU call 12: statistics epilogues of the transposed layers (own kernel build) and of the stem -- tests, step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "deconv or stats or conv2d" > $O/r4_tests12a.log 2>&1
tail -3 $O/r4_tests12a.log
grep -q "failed\|error" $O/r4_tests12a.log && exit 1
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/r4_all12.log 2>&1
tail -4 $O/r4_all12.log
for i in 1 2 3; do
  for bits in 0 1 2 3; do
    echo -n "stats epilogues (1 transposed, 2 stem) = $bits   "
    timeout -k 10 200 python tools/ab_deconv_stats.py $bits 2>/dev/null | tail -n 1
  done
done > $O/r4_ab_stats_epilogues.txt 2>&1
cat $O/r4_ab_stats_epilogues.txt
