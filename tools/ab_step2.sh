#!/bin/bash
# usage: tools/ab_step2.sh <variantA> <variantB> [rounds] -- the cycle step with two variant builds alternately on one box
cd "$(dirname "$0")/.."
n=${3:-3}
for i in $(seq 1 $n); do
  for v in $1 $2; do
    echo -n "$v  "
    timeout -k 10 200 python bench.py --lib sg-gan-tf2_amd/libsggan_$v.so --no-cpu-baseline --no-f32-leg --no-reference-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); ro=d.get('roofline_others',{}); print(round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms  roofline', round(d['roofline'].get('frac'),3), {k: round(v.get('frac'),3) for k,v in ro.items() if isinstance(v,dict) and v.get('frac')})" || exit 1
  done
done
