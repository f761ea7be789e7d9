#!/bin/bash
# Wave-state / instruction-mix counters of the residual-conv halo GEMM (run ON the GPU box): bash tools/pmc_waves.sh [op] [n]
# (separate --pmc passes, --kernel-trace only; the SQ cycle counters tick in units of 4 cycles)
R=${GRAFT_REPO_ROOT:-/root/repo}
OP=${1:-fwd}; NIMG=${2:-8}
O=$R/gpurun_out/pmc_waves
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum"; do
  d=$O/${OP}_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/bench_conv.py --n $NIMG --ops $OP --iters 6 > $d.log 2>&1 || echo "pass $c failed: $(tail -2 $d.log | tr '\n' ' ')"
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$O/${OP}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if any(k in n for k in ("conv3x3_halo_gemm","conv3x3_wgrad_halo")):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(f"{k:40s} launches {len(v):3d} mean {sum(v)/len(v):.4g}")
PY
rm -rf $O/${OP}_*/
