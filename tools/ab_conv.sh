#!/bin/bash
# A/B timing of library variants on one box, interleaved (box-to-box spread is +-5 %, so only same-call numbers compare):
#   bash tools/ab_conv.sh "<bench_conv args>" libsggan.so libsggan_x.so ...     (run ON the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="$1"; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib (rep $rep)"
    SGG_LIB_PATH=$R/sg-gan-tf2_amd/$lib python $R/tools/bench_conv.py $ARGS 2>&1 | grep -v amdgpu.ids
  done
done
