"""CPU-side checks of the drop-in boundary (no GPU, no compute calls): the C-ABI library loads and exports
every symbol include/sggan.h declares; argument validation that does not touch the device; host logic."""
import ctypes
import os
import re

import numpy as np
import pytest

import sggan_amd
from sggan_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sggan.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sgg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(A.LIB_PATH), "run `python __graft_entry__.py build` first"
    L = ctypes.CDLL(A.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/sggan.h but not exported by libsggan.so"
    assert set(A.SIGNATURES) == set(declared), set(A.SIGNATURES) ^ set(declared)


def test_version_and_strerror():
    L = A.lib()
    assert L.sgg_version() == 100
    assert L.sgg_strerror(0) == b"SGG_OK" and b"EINVAL" in L.sgg_strerror(-1) and b"WORKSPACE" in L.sgg_strerror(-4)


def test_host_side_validation_without_gpu():
    L = A.lib()
    bad = A.ConvDesc(1, 8, 8, 7, 8, 3, 3, 1, 0, 0, 6, 6, 0, 0)            # C not a multiple of 8
    assert L.sgg_conv2d_bwd_weight_workspace(ctypes.byref(bad)) == 0
    assert L.sgg_conv2d_fwd(ctypes.byref(bad), None, None, None, None, 0, 0.0, None, 0, None) == -1
    tail = A.ConvDesc(8, 7, 15, 512, 512, 3, 3, 1, 0, 0, 5, 13, 0, 1)            # D.h33: tiny output -> split-K slabs
    assert L.sgg_conv2d_fwd_workspace(ctypes.byref(tail)) > 0
    refl = A.ConvDesc(1, 2, 8, 8, 8, 7, 7, 1, 3, 3, 2, 8, 1, 0)           # REFLECT pad >= size
    assert L.sgg_conv2d_bwd_weight_workspace(ctypes.byref(refl)) == 0
    ok = A.ConvDesc(8, 64, 128, 256, 256, 3, 3, 1, 1, 1, 64, 128, 1, 1)    # the bench's residual conv
    ws = L.sgg_conv2d_bwd_weight_workspace(ctypes.byref(ok))
    assert ws > 0 and ws % (9 * 256 * 256 * 4) == 0
    assert L.sgg_instnorm_workspace(8, 64 * 128, 256) > 0 and L.sgg_instnorm_workspace(0, 1, 8) == 0
    assert L.sgg_adam(None, None, None, None, 10, 1, 1e-3, 0.5, 0.999, 1e-7, 1.0, None) == -1
    assert L.sgg_seg_class_map(None, 3, 0, None, None) == 0               # empty image is a no-op


def test_seg_class_table_matches_reference_map():
    """The kernel's colour table (host copy) is exactly segment_class.py:63-66."""
    from oracle import sggan_oracle as O
    import sggan_amd.segment_class as sc
    m = sc.cityscape()
    assert dict(m) == {k: v for k, v in O.CITYSCAPE_MAP}
    assert m[(1, 2, 3)] == 0 and sc.num_seg_masks == 8                     # defaultdict(int) semantics


def test_param_layout_and_geometry_helpers():
    from sggan_amd import kernels as K
    from sggan_amd.module import discriminator_param_specs, generator_param_specs
    from oracle import sggan_oracle as O
    assert [(n, tuple(s)) for n, s in generator_param_specs()] == [(n, tuple(s)) for n, s in O.generator_param_shapes()]
    assert [(n, tuple(s)) for n, s in discriminator_param_specs()] == [(n, tuple(s)) for n, s in O.discriminator_param_shapes()]
    assert K.same_pads(256, 3, 2) == (0, 1, 128) and K.same_pads(15, 3, 2) == (1, 1, 8) and K.cpad(34) == 40 and K.cpad(3) == 8


def test_product_path_has_no_cpu_fallback():
    """Networks refuse to run off-GPU, and nothing under the package imports the oracle."""
    import torch
    with pytest.raises(RuntimeError):
        sggan_amd.Generator(device="cpu")
    pkg = os.path.join(ROOT, "sg-gan-tf2_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, f)).read().replace("CPU oracle under oracle/ is test infrastructure", ""), f


def test_gradient_bucket_plan_covers_the_buffer_in_backward_order():
    """module.plan_buckets (the data-parallel step's layer-group buckets): contiguous, disjoint, covering the
    whole flat gradient buffer, cut at layer boundaries, the stem as a small bucket of its own."""
    from sggan_amd.module import ParamStore, generator_param_specs, plan_buckets
    P = ParamStore(generator_param_specs(), "cpu")
    names = ["c1", "c2", "c3"] + [f"r{i}{x}" for i in range(1, 10) for x in "ab"] + ["d1", "d2", "out"]
    assert plan_buckets(P, names, 1) == [("c1", 0, P.numel)]
    for n in (2, 3, 4, 6, 50):
        plan = plan_buckets(P, names, n)
        assert plan[0][:2] == ("c1", 0) and plan[-1][2] == P.numel and len(plan) <= len(names)
        assert all(a[2] == b[1] and a[1] < a[2] for a, b in zip(plan, plan[1:] + [(None, P.numel, None)]))
        assert all(lo == P.index[first + "_w"][0] for first, lo, _ in plan)
        if n <= 6:
            assert plan[1][0] == "r1a" and plan[0][2] - plan[0][1] < 0.05 * P.numel        # stem: c1..c3, 3 % of the buffer
            sizes = [hi - lo for _, lo, hi in plan[1:]]
            assert max(sizes) < 1.35 * min(sizes)


def test_merge_tiles_like_the_reference_loop():
    """utils.merge (utils.py:261-269) against the paste loop it replaces, incl. a partly filled grid."""
    from sggan_amd.utils import merge
    rng = np.random.default_rng(0)
    for b, size in ((6, (2, 3)), (4, (2, 3)), (1, (1, 1)), (8, (4, 2))):
        imgs = rng.integers(0, 256, (b, 5, 7, 3)).astype(np.uint8)
        h, w = 5, 7
        exp = np.zeros((h * size[0], w * size[1], 3))
        for idx, im in enumerate(imgs):
            i, j = idx % size[1], idx // size[1]
            exp[j * h:j * h + h, i * w:i * w + w, :] = im
        assert np.array_equal(merge(imgs, size), exp.astype(np.uint8))


def test_package_reads_no_environment():
    """neither the library nor the Python side of the drop-in takes configuration from the environment --
    switches are constructor arguments (default_args), the library path is explicit (_abi.use_library).  build.py (the
    build script, HIPCC override) is not part of the runtime path."""
    pkg = os.path.join(ROOT, "sg-gan-tf2_amd")
    for f in sorted(os.listdir(pkg)):
        if f.endswith(".py") and f != "build.py":
            src = open(os.path.join(pkg, f)).read()
            assert "os.environ" not in src and "getenv" not in src, f


def test_merge_rejects_a_batch_that_does_not_fit_the_grid():
    """utils.merge (utils.py:261-269): the reference's paste loop fails on the first image outside the size[0] x size[1] grid;
    the vectorised form must not silently drop images."""
    import numpy as np
    from sggan_amd.utils import merge
    imgs = np.arange(3 * 2 * 2 * 3, dtype=np.float64).reshape(3, 2, 2, 3)
    out = merge(imgs, [2, 2])
    assert out.shape == (4, 4, 3) and np.array_equal(out[:2, 2:], imgs[1].astype(np.uint8)) and not out[2:, 2:].any()
    with pytest.raises(ValueError):
        merge(imgs, [1, 2])
