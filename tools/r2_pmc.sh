#!/bin/bash
# All PMC passes of the round (run ON the GPU box): bash tools/r2_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/pmc_traffic.sh > $R/gpurun_out/r2_pmc_traffic.log 2>&1
bash $R/tools/pmc_lds.sh > $R/gpurun_out/r2_pmc_lds.log 2>&1
bash $R/tools/pmc_in.sh > $R/gpurun_out/r2_pmc_in.log 2>&1
tail -12 $R/gpurun_out/r2_pmc_lds.log
