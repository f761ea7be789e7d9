#!/bin/bash
# round-4 GPU call 14: streaming LDS-DMA of once-read GEMM operands (weight-gradient x / dy, halo rows), step-level A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for i in 1 2 3; do
  for lib in libsggan.so libsggan_w9nt1.so libsggan_w9nt3.so libsggan_h3nt3.so; do
    echo -n "$lib  "
    timeout -k 10 200 python bench.py --lib sg-gan-tf2_amd/$lib --no-cpu-baseline --no-f32-leg --no-reference-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); ro=d.get('roofline_others',{}); print(round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms  roofline', d['roofline']['name'], round(d['roofline'].get('frac'),3), {k: round(v.get('frac'),3) for k,v in ro.items() if isinstance(v,dict) and v.get('frac')})" || exit 1
  done
done > $O/r4_ab_nt_dma.txt 2>&1
cat $O/r4_ab_nt_dma.txt
