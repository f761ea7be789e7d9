#!/bin/bash
# LDS bank-conflict / MFMA-busy counters of the three residual-conv kernels (run ON the GPU box): bash tools/pmc_lds.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_lds
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for op in fwd dgrad wgrad; do
  for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
    d=$O/${op}_$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/bench_conv.py --ops $op --iters 6 > $d.log 2>&1 || echo "pass $op $c failed"
  done
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    op=f.split("pmc_lds/")[1].split("_")[0]
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if any(k in n for k in ("conv3x3_halo_gemm","conv3x3_wgrad_halo")):
            agg[(op, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, "launches", len(v), "mean %.4g" % (sum(v)/len(v)))
PY
