#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for ab in 0 1 2 3; do
  echo "== SGG_ABLATE=$ab"
  SGG_ABLATE=$ab SGG_LIB_PATH=$R/sg-gan-tf2_amd/libsggan_lab.so python $R/tools/bench_conv.py --n 8 --h 256 --w 512 --c 64 --k 3 --r 7 --pad REFLECT-3 --iters 30 --ops wgrad 2>&1 | grep wgrad
done
