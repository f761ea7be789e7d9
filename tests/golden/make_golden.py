#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Run in the AUTHORING container only (needs /root/reference for the two gta
label pairs; the GPU box never sees the reference):

    python tests/golden/make_golden.py

Writes
  segclass_gta.npz   -- PINNED known-answer data for segment_class.py:60-99: the two
                        input/output pairs the reference ships
                        (datasets/gta/trainA_seg/00005.png -> trainA_seg_class/00005.png,
                         datasets/gta/trainB_seg/aachen_000000_000019.png -> trainB_seg_class/...)
                        as 64x128 crops that between them cover every class that occurs,
                        the whole-image class histograms, and a CRC32 of each whole expected
                        image.  Data only: RGB inputs and the reference's own expected outputs.
  oracle_small.npz   -- oracle-generated (PARITY UNPINNED) float vectors for a reduced
                        generator/discriminator (gf_dim=df_dim=8, 2 res blocks): parameters,
                        inputs, forward outputs, losses, gradients, post-Adam parameters.
  oracle_full.json   -- oracle-generated checksums for the full-size networks at N=2,
                        128x128 (seeded parameters; losses, gradient norms, output stats).
  mask_zoom_city.npz -- known-answer data for utils.py:158-165 + 197-199 (one_hot + scipy.ndimage.zoom): three
                        of the reference's own class-index maps (datasets/city/trainA_seg_class/*.png, 1024x2048,
                        34 labelIds) and what ``scipy.ndimage.zoom(one_hot(idx), (H/34/h, W/34/w, 1), mode="nearest")``
                        returns for them at the 128x128 (4x4) and 512x256 (8x15) loader grids -- the reference's own
                        call on the reference's own data, evaluated with the scipy installed here (1.15.x; the
                        reference pinned 1.4.1 -- spline boundary handling for mode='nearest' changed in 1.6).
  oracle_d256.npz    -- oracle-generated (PARITY UNPINNED) reference-mode step at 1x256x256 with the FULL-WIDTH
                        discriminator (df_dim=64: 512-channel tail, D map 5x5, so every D gradient is non-zero) and
                        a reduced generator (gf_dim=8, 2 blocks).  D's 8.79 M parameters are re-generated from the
                        seed; its gradients / post-Adam parameters are stored as per-tensor norms, 4 random-sign
                        projections and a strided sample of 4096 entries.
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import sggan_oracle as O  # noqa: E402

REF = "/root/reference/datasets/gta"
PAIRS = (("trainA", "00005.png"), ("trainB", "aachen_000000_000019.png"))
CH, CW = 64, 128


def pick_crops(exp, want):
    """Greedy: 64x128 windows (on a 32-px grid) until every class in `want` is covered."""
    H, W = exp.shape
    crops, covered = [], set()
    cands = [(y, x) for y in range(0, H - CH + 1, 32) for x in range(0, W - CW + 1, 32)]
    sets = {c: set(np.unique(exp[c[0]:c[0] + CH, c[1]:c[1] + CW]).tolist()) for c in cands}
    while covered != want:
        best = max(cands, key=lambda c: (len(sets[c] - covered), len(sets[c])))
        if not sets[best] - covered:
            break
        crops.append(best)
        covered |= sets[best]
    return crops


def make_segclass():
    from PIL import Image
    out = {}
    for d, f in PAIRS:
        rgb = np.array(Image.open(f"{REF}/{d}_seg/{f}").convert("RGB"))       # 'P'/'RGBA' -> RGB
        rgba = np.array(Image.open(f"{REF}/{d}_seg/{f}").convert("RGBA"))
        exp = np.array(Image.open(f"{REF}/{d}_seg_class/{f}"))
        assert exp.dtype == np.uint8 and exp.shape == rgb.shape[:2]
        got = O.seg_class_map(rgb)
        assert np.array_equal(got, exp), f"oracle disagrees with reference fixture {d}/{f}"
        want = set(np.unique(exp).tolist())
        crops = pick_crops(exp, want)
        crng = np.random.default_rng(5)                      # + a few seeded windows (class borders)
        crops += [(int(crng.integers(0, exp.shape[0] - CH)), int(crng.integers(0, exp.shape[1] - CW))) for _ in range(5)]
        out[f"{d}_hist"] = np.bincount(exp.ravel(), minlength=8).astype(np.int64)
        out[f"{d}_shape"] = np.array(exp.shape)
        out[f"{d}_crc32_expected"] = np.array([zlib.crc32(exp.tobytes())], np.uint32)
        out[f"{d}_crc32_rgb"] = np.array([zlib.crc32(rgb.tobytes())], np.uint32)
        out[f"{d}_crop_yx"] = np.array(crops)
        out[f"{d}_crop_rgb"] = np.stack([rgb[y:y + CH, x:x + CW] for y, x in crops])
        out[f"{d}_crop_rgba"] = np.stack([rgba[y:y + CH, x:x + CW] for y, x in crops])
        out[f"{d}_crop_expected"] = np.stack([exp[y:y + CH, x:x + CW] for y, x in crops])
        print(d, f, "classes", sorted(want), "crops", crops)
    np.savez_compressed(os.path.join(HERE, "segclass_gta.npz"), **out)


def make_oracle_small():
    rng = np.random.default_rng(19)
    gsh = O.generator_param_shapes(gf_dim=8, n_blocks=2)
    dsh = O.discriminator_param_shapes(df_dim=8, segment_class=34)
    PG = O.init_params(gsh, rng, perturb=0.1)
    PD = O.init_params(dsh, rng, perturb=0.1)
    N, H, W = 2, 128, 128
    # images as the loader delivers them: 8-bit levels / 255 (utils.py:195-196 resize of uint8 PNGs)
    real_u8 = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    seg_u8 = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    idx = rng.integers(0, 34, (N, 4, 4))
    mask = np.stack([O.one_hot(i, 34) for i in idx]).astype(np.float64)
    out = {"real_A_u8": real_u8, "seg_A_u8": seg_u8, "mask_idx": idx.astype(np.uint8)}
    for k, v in PG.items():
        out["PG/" + k] = v.astype(np.float32)
    for k, v in PD.items():
        out["PD/" + k] = v.astype(np.float32)
    # fixture inputs are float32-representable; expected outputs are the float64 oracle on exactly those
    PG32 = {k: v.astype(np.float32).astype(np.float64) for k, v in PG.items()}
    PD32 = {k: v.astype(np.float32).astype(np.float64) for k, v in PD.items()}
    real = (real_u8.astype(np.float32) / np.float32(255)).astype(np.float64)
    seg = (seg_u8.astype(np.float32) / np.float32(255)).astype(np.float64)
    r = O.train_step(PG32, PD32, real, seg, mask, n_blocks=2)
    out.update({"fake_A": r["fake_A"].astype(np.float32), "da_real": r["da_real"], "da_fake": r["da_fake"],
                "gen_loss": np.array(r["gen_loss"]), "disc_loss": np.array(r["disc_loss"])})
    for k in PG:
        out["gG/" + k] = r["gG"][k].astype(np.float32)
        out["newPG/" + k] = r["PG"][k].astype(np.float32)
    for k in PD:
        out["gD/" + k] = r["gD"][k].astype(np.float32)
        out["newPD/" + k] = r["PD"][k].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), **out)
    print("small step: gen_loss", r["gen_loss"], "disc_loss", r["disc_loss"])


def full_inputs(seed=19, N=2, H=128, W=128):
    """Seeded full-size parameters + inputs (float32-representable), shared with the tests."""
    rng = np.random.default_rng(seed)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    PG = {k: f32(v) for k, v in O.init_params(O.generator_param_shapes(), rng, perturb=0.1).items()}
    PD = {k: f32(v) for k, v in O.init_params(O.discriminator_param_shapes(), rng, perturb=0.1).items()}
    real = f32(rng.uniform(0, 1, (N, H, W, 3)))
    seg = f32(rng.uniform(0, 1, (N, H, W, 3)))
    idx = rng.integers(0, 34, (N, 4, 4))
    mask = np.stack([O.one_hot(i, 34) for i in idx]).astype(np.float64)
    return PG, PD, real, seg, mask


def make_oracle_full():
    PG, PD, real, seg, mask = full_inputs()
    r = O.train_step(PG, PD, real, seg, mask)
    norm = lambda d: {k: float(np.sqrt((v ** 2).sum())) for k, v in d.items()}
    js = {"seed": 19, "N": 2, "H": 128, "W": 128,
          "gen_loss": r["gen_loss"], "disc_loss": r["disc_loss"],
          "fake_A_mean": float(r["fake_A"].mean()), "fake_A_abs_mean": float(np.abs(r["fake_A"]).mean()),
          "fake_A_first8": r["fake_A"].ravel()[:8].tolist(),
          "da_real": r["da_real"].ravel().tolist(), "da_fake": r["da_fake"].ravel().tolist(),
          "gG_norm": norm(r["gG"]), "gD_norm": norm(r["gD"]),
          "newPG_norm": norm(r["PG"]), "newPD_norm": norm(r["PD"]),
          "param_crc32": {"PG": zlib.crc32(b"".join(v.astype(np.float32).tobytes() for v in PG.values())),
                          "PD": zlib.crc32(b"".join(v.astype(np.float32).tobytes() for v in PD.values()))}}
    with open(os.path.join(HERE, "oracle_full.json"), "w") as f:
        json.dump(js, f, indent=1)
    print("full step: gen_loss", r["gen_loss"], "disc_loss", r["disc_loss"])


CITY = "/root/reference/datasets/city/trainA_seg_class"
CITY_MAPS = ("aachen_000000.png", "aachen_000059.png", "aachen_000168.png")
ZOOM_SIZES = ((128, 128), (256, 512))              # (image_height, image_width) -> 4x4 and 8x15 mask grids


def make_mask_zoom():
    from PIL import Image
    out = {"names": np.array(CITY_MAPS), "sizes": np.array(ZOOM_SIZES)}
    for i, f in enumerate(CITY_MAPS):
        idx = np.array(Image.open(f"{CITY}/{f}"))
        assert idx.dtype == np.uint8 and idx.ndim == 2 and idx.max() < 34
        out[f"idx{i}"] = idx
        hot = O.one_hot(idx, 34)                                       # utils.py:158-165
        for (H, W) in ZOOM_SIZES:
            z = O.zoom_mask_reference(hot, H, W)                       # utils.py:197-199, scipy as installed
            assert z.shape == (round(H / 34), round(W / 34), 34) and set(np.unique(z).tolist()) <= {0, 1}
            mine = O.mask_from_index(idx, 34, z.shape[0], z.shape[1])
            print(f, (H, W), "grid", z.shape[:2], "cells where the nearest-index rule differs from zoom:", int((mine != z).any(-1).sum()))
            out[f"zoom{i}_{H}x{W}"] = z.astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "mask_zoom_city.npz"), **out)


SIGNS = 4
SAMPLE = 4096


def tensor_digest(v):
    """norm + SIGNS random-sign projections + a strided sample: enough to pin a multi-million-entry tensor in a small file."""
    v = np.asarray(v, np.float64).ravel()
    proj = []
    for j in range(SIGNS):
        s = np.random.default_rng(1000 + j).integers(0, 2, v.size).astype(np.float64) * 2 - 1
        proj.append(float((v * s).sum()))
    step = max(1, v.size // SAMPLE)
    return np.array([np.sqrt((v * v).sum())] + proj), v[::step][:SAMPLE].astype(np.float32)


def d256_inputs(seed=23):
    """Seeded parameters + inputs of the oracle_d256 fixture (float32-representable), shared with the tests."""
    rng = np.random.default_rng(seed)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    PG = {k: f32(v) for k, v in O.init_params(O.generator_param_shapes(gf_dim=8, n_blocks=2), rng, perturb=0.1).items()}
    PD = {k: f32(v) for k, v in O.init_params(O.discriminator_param_shapes(df_dim=64), rng, perturb=0.1).items()}
    N, H, W = 1, 256, 256
    real = f32(rng.integers(0, 256, (N, H, W, 3)) / 255.0)
    seg = f32(rng.integers(0, 256, (N, H, W, 3)) / 255.0)
    idx = rng.integers(0, 34, (N, 5, 5))
    mask = np.stack([O.one_hot(i, 34) for i in idx]).astype(np.float64)
    return PG, PD, real, seg, mask


def make_oracle_d256():
    PG, PD, real, seg, mask = d256_inputs()
    r = O.train_step(PG, PD, real, seg, mask, n_blocks=2)
    out = {"seed": np.array(23), "gen_loss": np.array(r["gen_loss"]), "disc_loss": np.array(r["disc_loss"]),
           "fake_A": r["fake_A"].astype(np.float32), "da_real": r["da_real"], "da_fake": r["da_fake"],
           "PD_crc32": np.array([zlib.crc32(b"".join(v.astype(np.float32).tobytes() for v in PD.values()))], np.uint32)}
    for k in PG:
        out["gG/" + k] = r["gG"][k].astype(np.float32)
        out["newPG/" + k] = r["PG"][k].astype(np.float32)
    # the same step by the independent PyTorch-CPU composition: float64 must agree with the NumPy oracle to rounding;
    # its float32 run gives the inherent f32-vs-f64 distance per gradient tensor ("f32_floor", see the test)
    import torch
    from oracle import torch_restatement as T
    t64 = T.RefStep(PG, PD, torch.float64, n_blocks=2).step(real, seg, mask, apply=False)
    t32 = T.RefStep(PG, PD, torch.float32, n_blocks=2).step(real, seg, mask, apply=False)
    for k in PD:
        a, b, c = r["gD"][k], t64["gD"][k].numpy(), t32["gD"][k].numpy().astype(np.float64)
        n = max(np.sqrt((a * a).sum()), 1e-30)
        assert np.sqrt(((a - b) ** 2).sum()) < 1e-9 * n + 1e-12, ("oracle vs torch f64", k)
        out["f32_floor/" + k] = np.array(max(np.sqrt(((a - c) ** 2).sum()) / n, np.abs(a - c).max() / max(np.abs(a).max(), 1e-30)))
    for k in PD:
        out["gD_digest/" + k], out["gD_sample/" + k] = tensor_digest(r["gD"][k])
        out["newPD_digest/" + k], out["newPD_sample/" + k] = tensor_digest(r["PD"][k])
        print("gD", k, "norm %.4e" % out["gD_digest/" + k][0])
    np.savez_compressed(os.path.join(HERE, "oracle_d256.npz"), **out)
    print("d256 step: gen_loss", r["gen_loss"], "disc_loss", r["disc_loss"])


def bf16_mid_inputs(seed=29):
    """Seeded parameters + inputs of the 'mid' case of oracle_bf16_emulated.npz: gf_dim = df_dim = 16, 3 residual blocks,
    1x256x256 (D map 5x5: the generator receives D's gradient and every D gradient is non-zero)."""
    rng = np.random.default_rng(seed)
    f32 = lambda a: a.astype(np.float32)
    PG = {k: f32(v) for k, v in O.init_params(O.generator_param_shapes(gf_dim=16, n_blocks=3), rng, perturb=0.1).items()}
    PD = {k: f32(v) for k, v in O.init_params(O.discriminator_param_shapes(df_dim=16), rng, perturb=0.1).items()}
    real = f32(rng.integers(0, 256, (1, 256, 256, 3)) / 255.0)
    seg = f32(rng.integers(0, 256, (1, 256, 256, 3)) / 255.0)
    mask = np.stack([O.one_hot(i, 34) for i in rng.integers(0, 34, (1, 5, 5))]).astype(np.float32)
    return PG, PD, real, seg, mask


def small_inputs():
    z = np.load(os.path.join(HERE, "oracle_small.npz"))
    PG = {k[3:]: z[k] for k in z.files if k.startswith("PG/")}
    PD = {k[3:]: z[k] for k in z.files if k.startswith("PD/")}
    real = z["real_A_u8"].astype(np.float32) / np.float32(255)
    seg = z["seg_A_u8"].astype(np.float32) / np.float32(255)
    mask = np.stack([O.one_hot(i, 34) for i in z["mask_idx"]]).astype(np.float32)
    return PG, PD, real, seg, mask


def make_bf16_emulated():
    """oracle_bf16_emulated.npz -- the step-level oracle of the TIMED (bf16-storage) path: the PyTorch-CPU restatement of the
    reference-mode step evaluated in float64 with every tensor the MI355X path stores rounded to bfloat16 exactly where it
    stores one (oracle/torch_restatement.py bf16_storage: conv outputs, layer outputs, inputs, the GEMM's weight copy; values
    on the way forward, gradients on the way back).  (The rounded 29-layer chain is sensitive: a value that crosses a rounding
    boundary moves the values behind it by far more than f32 noise, which moves further ones -- by the generator's output
    about a quarter of the pixels differ by one bf16 step between two correct evaluations, `fake_floor`.)  Two cases: 'small' (the oracle_small.npz inputs: gf = df = 8, 2 blocks,
    2x128x128) and 'mid' (bf16_mid_inputs).  Per case: losses, the generator image, every parameter gradient, and
    ``floor/<net>/<tensor>`` = 1 - cosine between this float64 evaluation and the SAME emulation evaluated in float32 -- two
    correct implementations of one storage policy differ by that much, because f32-vs-f64 summation noise moves values across
    bf16 rounding boundaries (most easily the small ones) and the rounded backward chain carries each flip on."""
    import torch
    from oracle import torch_restatement as T
    out = {}
    for case, (PG, PD, real, seg, mask), nb in (("small", small_inputs(), 2), ("mid", bf16_mid_inputs(), 3)):
        res = {}
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            with T.bf16_storage():
                res[tag] = T.RefStep(PG, PD, dtype, n_blocks=nb).step(real, seg, mask, apply=False)
        plain = T.RefStep(PG, PD, torch.float64, n_blocks=nb).step(real, seg, mask, apply=False)
        r = res["f64"]
        out[f"{case}/gen_loss"], out[f"{case}/disc_loss"] = np.array(r["gen_loss"]), np.array(r["disc_loss"])
        out[f"{case}/gen_loss_f32path"], out[f"{case}/disc_loss_f32path"] = np.array(plain["gen_loss"]), np.array(plain["disc_loss"])
        out[f"{case}/fake_A"] = r["fake_A"].numpy().astype(np.float32)
        fa, fb = r["fake_A"].numpy().astype(np.float64), res["f32"]["fake_A"].numpy().astype(np.float64)
        d, step = np.abs(fa - fb), np.maximum(np.abs(fa), 2.0 ** -7) * 2.0 ** -7         # one bf16 step at that magnitude
        out[f"{case}/fake_floor"] = np.array([d.mean(), d.max(), (d > 1.01 * step).mean()])   # same emulation, f32 vs f64
        out[f"{case}/da_fake"], out[f"{case}/da_real"] = r["da_fake"].numpy(), r["da_real"].numpy()
        cosd = lambda a, b: 1.0 - float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm() + 1e-300))
        worst = 0.0
        for net in ("gG", "gD"):
            for k, g in r[net].items():
                out[f"{case}/{net}/{k}"] = g.numpy().astype(np.float32)
                fl = cosd(g, res["f32"][net][k]) if float(g.double().norm()) > 0 else 0.0
                out[f"{case}/floor/{net}/{k}"] = np.array(fl)
                out[f"{case}/f32path_dist/{net}/{k}"] = np.array(cosd(g, plain[net][k]) if float(g.double().norm()) > 0 else 0.0)
                worst = max(worst, fl)
        print(case, "gen_loss", r["gen_loss"], "(f32 path %.6f)" % plain["gen_loss"], "disc_loss", r["disc_loss"],
              "worst floor 1-cos %.2e" % worst)
    np.savez_compressed(os.path.join(HERE, "oracle_bf16_emulated.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["segclass", "small", "full", "zoom", "d256", "bf16emu"]
    for name, fn in (("segclass", make_segclass), ("small", make_oracle_small), ("full", make_oracle_full),
                     ("zoom", make_mask_zoom), ("d256", make_oracle_d256), ("bf16emu", make_bf16_emulated)):
        if name in which:
            fn()
