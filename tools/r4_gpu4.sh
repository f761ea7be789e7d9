#!/bin/bash
# round-4 GPU call 4: persistent halo GEMM -- tests, A/B against the one-tile-per-block form, LDS bank-conflict counters of the patch layouts
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/r4_all4.log 2>&1
tail -4 $O/r4_all4.log
grep -q " passed" $O/r4_all4.log || exit 1
grep -q "failed" $O/r4_all4.log && exit 1
bash tools/ab_conv.sh "--n 16 --iters 60 --ops fwd_pair,dgrad_pair" libsggan.so libsggan_np.so libsggan_pw16.so libsggan_pp0.so libsggan_pw16p0.so libsggan_w1p0.so > $O/r4_ab_persist.txt 2>&1
grep -v "^$" $O/r4_ab_persist.txt | tail -40
cd /tmp && export TMPDIR=/tmp
for lib in np w1p0; do
  for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    d=$O/pmc4/${lib}_$(echo $c | tr ' ' '_')
    mkdir -p $d
    SGG_LIB_PATH=$R/sg-gan-tf2_amd/libsggan_$lib.so rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/bench_conv.py --n 16 --ops dgrad_pair --iters 6 > $d.log 2>&1 || echo "pass $lib $c failed"
  done
done
python3 - <<PY > $O/r4_pmc_patch_layout.txt
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$O/pmc4/**/*counter_collection.csv", recursive=True):
    lib=f.split("pmc4/")[1].split("_")[0]
    for r in csv.DictReader(open(f)):
        if "conv3x3_halo_gemm" in r["Kernel_Name"]:
            agg[(lib, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, "launches", len(v), "mean %.5g" % (sum(v)/len(v)))
PY
cat $O/r4_pmc_patch_layout.txt
