#!/bin/bash
# A/B of the normalise-on-load forward: where in the step the row transform runs (H3_NORM_AT variants) vs the separate apply pass
# build first:  for n in 0 2 3; do python sg-gan-tf2_amd/build.py --variant na$n -DH3_NORM_AT=$n; done
cd "$(dirname "$0")/.."
OPS=fwd_pair,dgrad_pair,in_apply_pair,fwd_normload_pair
for v in "" na0 na2 na3 ""; do
  if [ -z "$v" ]; then lib=sg-gan-tf2_amd/libsggan.so; else lib=sg-gan-tf2_amd/libsggan_$v.so; fi
  ops=$OPS; [ "$v" = base ] && ops=fwd_pair,dgrad_pair
  echo "== ${v:-default}"
  SGG_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/bench_conv.py --n 16 --iters 50 --rounds 7 --ops $ops || exit 1
done
