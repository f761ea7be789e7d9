#!/bin/bash
# round-4 GPU call 5: 16-byte stores in the stride-2 halo / generic GEMM epilogues -- tests, step A/B; final persistent form for the record
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/r4_all5.log 2>&1
tail -4 $O/r4_all5.log
grep -q " passed" $O/r4_all5.log || exit 1
grep -q "failed" $O/r4_all5.log && exit 1
bash tools/ab_conv.sh "--n 16 --iters 60 --ops fwd_pair,dgrad_pair" libsggan.so libsggan_pf.so > $O/r4_ab_persist_final.txt 2>&1
grep -v "^$" $O/r4_ab_persist_final.txt | tail -20
# step level: no wide stores anywhere (nw) / halo GEMM only (h3w) / all three kernels (libsggan.so)
for i in 1 2 3; do
  for lib in libsggan_nw.so libsggan_h3w.so libsggan.so; do
    echo -n "$lib  "
    timeout -k 10 200 python bench.py --lib sg-gan-tf2_amd/$lib --no-cpu-baseline --no-f32-leg --no-reference-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); ro=d.get('roofline_others',{}); print(round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms  roofline', d['roofline']['name'], round(d['roofline'].get('frac'),3), {k: round(v.get('frac'),3) for k,v in ro.items() if isinstance(v,dict) and v.get('frac')})" || exit 1
  done
done > $O/r4_ab_step_wide.txt 2>&1
cat $O/r4_ab_step_wide.txt
