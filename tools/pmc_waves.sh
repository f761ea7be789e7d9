#!/bin/bash
# Where the waves of the residual-conv forward kernel spend their cycles (run ON the GPU box): bash tools/pmc_waves.sh [op]
R=${GRAFT_REPO_ROOT:-/root/repo}
OP=${1:-fwd}
O=$R/gpurun_out/pmc_waves_$OP
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/bench_conv.py --ops $OP --iters 6 > $O/p$i.log 2>&1 || echo "pass $c failed"
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if any(k in n for k in ("conv3x3_halo_gemm","conv3x3_wgrad_halo")):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(f"{k:34s} launches {len(v):3d} mean {sum(v)/len(v):.4g}")
PY
