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


if __name__ == "__main__":
    make_segclass()
    make_oracle_small()
    make_oracle_full()
