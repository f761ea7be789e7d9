#!/bin/bash
# HBM traffic of the instance-norm kernels on the residual tensor (N=8, 64x128, C=256, bf16) from rocprofv3 PMC passes
# (run ON the GPU box): bash tools/pmc_in.sh -> gpurun_out/pmc_in/in_traffic.json
# One counter set per run, --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section); FETCH_SIZE is doubled there.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_in
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dir in fwd bwd; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${dir}_$c -- python3 $R/tools/bench_in.py --only res --dir $dir > $O/${dir}_$c.log 2>&1 || echo "pass $dir $c failed"
  done
done
python3 - <<PY > $O/in_traffic.json
import csv, glob, json, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    d = "bwd" if "/bwd_" in f else "fwd"             # one direction per pass (tools/bench_in.py --dir)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("in_apply_kernel", "in_partial_kernel", "in_finalize"):
            if key in n:
                vals[key + "_" + d][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, m in sorted(vals.items()):
    mean = {c: sum(v) / len(v) for c, v in m.items()}
    e = {"launches_per_counter": {c: len(v) for c, v in m.items()}}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = mean["FETCH_SIZE"], mean["WRITE_SIZE"]
        e["hbm_bytes_per_launch"] = (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
    out[k] = e
out["note"] = ("rocprofv3 --pmc (separate passes, --kernel-trace only) on tools/bench_in.py --only res: tensor 8x64x128x256 bf16 = 33.55 MB; "
               "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); algorithmic bytes: partial fwd 1 read, apply fwd 1 read + 1 write, "
               "partial bwd 2 reads, apply bwd 2 reads + 1 write")
print(json.dumps(out, indent=1))
PY
cat $O/in_traffic.json
