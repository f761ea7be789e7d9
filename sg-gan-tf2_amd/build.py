"""Build libsggan.so (the C-ABI HIP library) in-tree for gfx950.

    python sg-gan-tf2_amd/build.py [--force] [--lab]

--lab builds libsggan_lab.so with -DSGG_LAB: the SGG_* kernel-selection / ablation environment switches and the
sgg_debug_* exports used by tools/ (point SGG_LIB_PATH at it).  The shipped libsggan.so reads no environment.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsggan.so")
SOURCES = ["conv.hip", "norm.hip", "misc.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-Wno-inline-asm"]


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


def build_lib(force=False, verbose=False, lab=False, variant=None, defines=()):
    """variant: build libsggan_<variant>.so with extra -D defines (A/B timing of kernel variants; SGG_LIB_PATH selects it)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sggan.h")]
    tag = variant or ("lab" if lab else None)
    OUT = os.path.join(HERE, f"libsggan_{tag}.so" if tag else "libsggan.so")
    if not force and _newer(OUT, deps):
        return OUT
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    objs = []
    flags = FLAGS + (["-DSGG_LAB"] if lab else []) + list(defines)

    def cc(src):
        obj = os.path.join(HERE, "build", src.replace(".hip", f".{tag}.o" if tag else ".o"))
        if force or not _newer(obj, deps):
            cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"hipcc failed on {src}")
            if verbose and r.stderr.strip():
                sys.stderr.write(r.stderr[-4000:])
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], check=True)
    return OUT


if __name__ == "__main__":
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print(build_lib(force="--force" in sys.argv, verbose=True, lab="--lab" in sys.argv, variant=var,
                    defines=[a for a in sys.argv[1:] if a.startswith("-D")]))
