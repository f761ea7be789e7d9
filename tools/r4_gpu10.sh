#!/bin/bash
# round-4 GPU call 10: more streaming loads (forward residual, slab reducer, data-gradient addend), step-level A/B; pool tests with the async staging
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_next_rows.py tests/test_gpu_exact.py -q -m gpu -x > $O/r4_tests10.log 2>&1
tail -3 $O/r4_tests10.log
for i in 1 2 3; do
  for lib in libsggan_nt0.so libsggan.so libsggan_nt15.so libsggan_nt7s.so libsggan_ntall.so; do
    echo -n "$lib  "
    timeout -k 10 200 python bench.py --lib sg-gan-tf2_amd/$lib --no-cpu-baseline --no-f32-leg --no-reference-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels'].get('res_instnorm_apply_fwd',{}); print(round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms  in_apply_fwd', round(k.get('avg_ms',0)*1e3,1), 'us  roofline', round(d['roofline']['frac'],3))" || exit 1
  done
done > $O/r4_ab_nt_loads2.txt 2>&1
cat $O/r4_ab_nt_loads2.txt
