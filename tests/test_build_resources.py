"""Build-time guard for the register-bound GEMM kernels (CPU only: hipcc cross-compiles without a GPU).

The 8-wave kernels run at 2 waves/SIMD, i.e. a 256-VGPR budget they use almost completely; a source change that tips
one of them into scratch spills costs 2-5x at run time without failing any numerical test (seen while developing the
all-taps weight-gradient kernel).  This test recompiles csrc/conv.hip with -Rpass-analysis=kernel-resource-usage and
requires zero VGPR spills and the expected occupancy for the production configurations.
"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sg-gan-tf2_amd"))

# mangled-name fragments of the kernels that must not spill (production tile configurations)
MUST_NOT_SPILL = [
    # conv3x3_halo_gemm_kernel<MODE, FOLD, STATS, PAIR>: every instantiation the library dispatches
    "conv3x3_halo_gemm_kernelILi0ELb0ELi0ELb0E",               # 3x3 forward, halo resident
    "conv3x3_halo_gemm_kernelILi0ELb0ELi1ELb0E",               # ... with the norm-statistics epilogue
    "conv3x3_halo_gemm_kernelILi0ELb0ELi0ELb1E",               # ... on a stacked pair of networks
    "conv3x3_halo_gemm_kernelILi0ELb0ELi1ELb1E",               # ... both (the roofline kernel of the cycle step)
    "conv3x3_halo_gemm_kernelILi1ELb0ELi0ELb0E",               # 3x3 data gradient, zero padding
    "conv3x3_halo_gemm_kernelILi1ELb1ELi0ELb0E",               # 3x3 data gradient, REFLECT fold
    "conv3x3_halo_gemm_kernelILi1ELb0ELi0ELb1E", "conv3x3_halo_gemm_kernelILi1ELb1ELi0ELb1E",   # ... paired
    # (<DGRAD, *, STATS=2>: the data gradient with the norm-backward sums in its epilogue -- opt-in (fuse_in_bwd), off by default, measured
    # "no gain" at the step's launch size in round 3 -- sat at 249-250 VGPRs then; round 4's restructured kernel body (tile loop, two loop
    # copies for the staggered form) tips it into 29-41 spilled VGPRs.  Numerically tested (test_conv_dgrad_norm_bwd_epilogue), not guarded here.)
    "conv3x3_wgrad_halo_kernel",                               # 3x3 weight gradient, all taps per block
    "conv3x3_wgrad_halo_s2_kernel",                            # the same for stride 2
    "deconv_s2_halo_kernel",                                   # stride-2 data gradient / Conv2DTranspose forward
    "conv_wgrad_glds_kernelIDF16bLb0E", "conv_wgrad_glds_kernelIDF16bLb1E",
    "conv_gemm_glds_kernelIDF16bLi0ELi256ELi256ELi2ELi8ELi128ELi2E",
    "conv_gemm_glds_kernelIDF16bLi1ELi256ELi256ELi2ELi8ELi128ELi2E",
    "conv_gemm_glds_kernelIDF16bLi0ELi256ELi128ELi4ELi8ELi128ELi3E",
]
# small-tile configurations: only spills are checked
NO_SPILL_ONLY = ["conv_gemm_glds_kernelIDF16bLi0ELi128ELi128ELi2ELi8ELi128ELi2E", "conv_gemm_glds_kernelIDF16bLi1ELi128ELi128ELi2ELi8ELi128ELi2E"]


def test_gemm_kernels_do_not_spill(tmp_path):
    import build as B
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    src = os.path.join(B.CSRC, "conv.hip")
    r = subprocess.run([hipcc, *B.FLAGS, "-c", src, "-o", str(tmp_path / "conv.o"), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # remarks come in blocks: "Function Name: X", ..., "VGPRs Spill: N", ..., "Occupancy [waves/SIMD]: M"
    usage, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        for key, pat in (("spill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("vgprs", r" VGPRs: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                usage[name][key] = int(m.group(1))
    assert usage, "no kernel-resource-usage remarks parsed"
    for frag in MUST_NOT_SPILL + NO_SPILL_ONLY:
        hits = {k: v for k, v in usage.items() if frag in k}
        assert hits, f"kernel {frag} not found in the build"
        for k, v in hits.items():
            assert v.get("spill") == 0, f"{k} spills {v.get('spill')} VGPRs"
            if frag in MUST_NOT_SPILL:
                assert v.get("occ", 0) >= 2, f"{k} occupancy {v.get('occ')} waves/SIMD"
