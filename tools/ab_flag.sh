#!/bin/bash
# usage: tools/ab_flag.sh "<flags A>" "<flags B>" [rounds] -- the cycle step with two bench.py flag sets alternately on one box
cd "$(dirname "$0")/.."
n=${3:-3}
for i in $(seq 1 $n); do
  for f in "$1" "$2"; do
    echo -n "[$f]  "
    timeout -k 10 200 python bench.py $f --no-cpu-baseline --no-f32-leg --no-reference-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms', 'gen_loss', d['gen_loss'], 'disc_loss', d['disc_loss'])" || exit 1
  done
done
