#!/bin/bash
# register / spill report of the halo GEMM instantiations for a set of -D defines (CPU, cross-compile): bash tools/kres.sh [-DX=1 ...]
R=$(cd $(dirname $0)/.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value "$@" -c $R/sg-gan-tf2_amd/csrc/conv.hip -o /tmp/kres_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name: _Z24${KRES_PAT:-conv3x3_halo}" | grep -E "Function Name| VGPRs:|VGPRs Spill" \
  | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - -
rm -f /tmp/kres_$$.o
