#!/bin/bash
# A/B of the 7x7 weight-gradient kernel (stem 3->64 and head 64->3 at 256x512, 8 images): variant build vs the in-tree library
cd "$(dirname "$0")/.."
for v in "$1" "" "$1" ""; do
  if [ -z "$v" ]; then lib=sg-gan-tf2_amd/libsggan.so; else lib=sg-gan-tf2_amd/libsggan_$v.so; fi
  echo "== ${v:-default}"
  SGG_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/bench_conv.py --n 8 --h 256 --w 512 --c 3 --k 64 --r 7 --pad REFLECT-3 --iters 30 --rounds 5 --ops fwd,dgrad,wgrad || exit 1
  SGG_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/bench_conv.py --n 8 --h 256 --w 512 --c 64 --k 3 --r 7 --pad REFLECT-3 --iters 30 --rounds 5 --ops fwd,dgrad,wgrad || exit 1
done
