"""Parse the rocprofv3 --pmc passes written by tools/pmc_traffic.sh into per-kernel HBM bytes per launch.

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, so it is
doubled (MI355X_MICROARCH.md, HBM section).  Only the main GEMM kernel of each op is counted (the data gradient's
small side-tensor gather and the weight gradient's slab reducer are listed separately).
"""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
OPS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fwd", "dgrad", "wgrad"]
NIMG = int(sys.argv[3]) if len(sys.argv) > 3 else 8
_G = ("conv3x3_halo_gemm", "conv_gemm_glds")
_W = ("conv3x3_wgrad_halo", "conv_wgrad_glds")
MAIN = {"fwd": _G, "dgrad": _G, "wgrad": _W, "fwd_pair": _G, "dgrad_pair": _G, "wgrad_pair2": _W}
AUX = {"dgrad": ("fold_halo_gather", "fold_gather"), "wgrad": ("wgrad_reduce",), "dgrad_pair": ("fold_halo_gather", "fold_gather"), "wgrad_pair2": ("wgrad_reduce",)}
GF8 = 77.309411328                                    # GFLOP of one 8-image residual conv
GFLOP = {"wgrad_pair2": 2 * NIMG / 8 * GF8}           # two networks x two applications of NIMG/2 images each
out = {}
for op in OPS:
    vals, aux = defaultdict(list), defaultdict(list)
    for f in glob.glob(os.path.join(root, op + "_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if any(m in name for m in MAIN[op]):
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            elif any(m in name for m in AUX.get(op, ())):
                aux[r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = lambda v: sum(v) / len(v) if v else None
    m = {k: mean(v) for k, v in vals.items()}
    e = {"launches_per_counter": {k: len(v) for k, v in vals.items()}}
    if m.get("FETCH_SIZE") is not None and m.get("WRITE_SIZE") is not None:
        e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = m["FETCH_SIZE"], m["WRITE_SIZE"]
        e["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None:
        e["TCC_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if aux:
        a = {k: mean(v) for k, v in aux.items()}
        if a.get("FETCH_SIZE") is not None and a.get("WRITE_SIZE") is not None:
            e["aux_kernel_hbm_bytes_per_launch"] = (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0
    e["gflop_per_launch"] = GFLOP.get(op, NIMG / 8 * GF8)
    e["note"] = (f"rocprofv3 --pmc (separate passes, --kernel-trace only) on tools/bench_conv.py --ops {op} --n {NIMG}, 64x128 C=K=256 3x3; FETCH_SIZE doubled "
                 "(gfx950 counts 128-B requests as 64 B); algorithmic bytes per 8 images and application: 33.6 MB in + 33.6 MB out (forward / data "
                 "gradient; + 33.6 MB for the skip-gradient addend in dgrad_pair) or 2 x 33.6 MB in (weight gradient), + 1.2 MB of weights per network")
    out["res_conv_" + op] = e
print(json.dumps(out, indent=1))
