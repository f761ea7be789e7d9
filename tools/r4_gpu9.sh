#!/bin/bash
# round-4 GPU call 9: streaming (nt) loads in the instance-norm apply passes, step-level A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for i in 1 2 3; do
  for lib in libsggan.so libsggan_nt1.so libsggan_nt6.so libsggan_nt7.so; do
    echo -n "$lib  "
    timeout -k 10 200 python bench.py --lib sg-gan-tf2_amd/$lib --no-cpu-baseline --no-f32-leg --no-reference-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels'].get('res_instnorm_apply_fwd',{}); print(round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms  in_apply_fwd', round(k.get('avg_ms',0)*1e3,1), 'us', round(k.get('gbs',0)), 'GB/s')" || exit 1
  done
done > $O/r4_ab_nt_loads.txt 2>&1
cat $O/r4_ab_nt_loads.txt
