#!/bin/bash
# round-4 GPU call 20: issuer waves' first fragment reads ahead of their DMA issue -- exactness, kernel and step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_exact.py tests/test_gpu_ops.py -q -m gpu -x > $O/r4_tests20.log 2>&1
tail -3 $O/r4_tests20.log
grep -q "failed\|rror" $O/r4_tests20.log && exit 1
bash tools/ab_conv.sh "--n 16 --iters 60 --ops fwd_pair,dgrad_pair" libsggan.so libsggan_ef0.so > $O/r4_ab_early_frags.txt 2>&1
grep -v "^$" $O/r4_ab_early_frags.txt | tail -20
for i in 1 2 3; do
  for lib in libsggan_ef0.so libsggan.so; do
    echo -n "$lib  "
    timeout -k 10 200 python bench.py --lib sg-gan-tf2_amd/$lib --no-cpu-baseline --no-f32-leg --no-reference-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); ro=d.get('roofline_others',{}); print(round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms  roofline', d['roofline']['name'], round(d['roofline'].get('frac'),3), {k: round(v.get('frac'),3) for k,v in ro.items() if isinstance(v,dict) and v.get('frac')})" || exit 1
  done
done > $O/r4_ab_early_frags_step.txt 2>&1
cat $O/r4_ab_early_frags_step.txt
