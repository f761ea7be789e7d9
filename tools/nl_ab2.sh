#!/bin/bash
# where the normalise-on-load forward loses its time: ablation builds (H3_NORM_ABL bits; results are wrong, timings only)
# build first:  for v in 1 2 4 7; do python sg-gan-tf2_amd/build.py --variant nabl$v -DH3_NORM_ABL=$v; done
cd "$(dirname "$0")/.."
for v in "" nabl1 nabl2 nabl4 nabl7; do
  if [ -z "$v" ]; then lib=sg-gan-tf2_amd/libsggan.so; else lib=sg-gan-tf2_amd/libsggan_$v.so; fi
  echo "== ${v:-default}"
  SGG_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/bench_conv.py --n 16 --iters 50 --rounds 5 --ops fwd_pair,fwd_normload_pair || exit 1
done
