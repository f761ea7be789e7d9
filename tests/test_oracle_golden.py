"""Oracle vs the committed fixtures: the PINNED segment_class known answers
(reference data) and the oracle's own regression vectors."""
import json
import os
import zlib

import numpy as np

from oracle import sggan_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def test_segclass_reference_pairs():
    z = np.load(os.path.join(G, "segclass_gta.npz"))
    seen = set()
    for d in ("trainA", "trainB"):
        rgb, rgba, exp = z[f"{d}_crop_rgb"], z[f"{d}_crop_rgba"], z[f"{d}_crop_expected"]
        for i in range(len(rgb)):
            assert np.array_equal(O.seg_class_map(rgb[i]), exp[i])
            assert np.array_equal(O.seg_class_map(rgba[i]), exp[i])      # alpha ignored: img[x,y,:3]
            seen |= set(np.unique(exp[i]).tolist())
        assert np.array_equal(O.seg_class_map_loop(rgb[0][:16, :32]), exp[0][:16, :32])
        assert int(z[f"{d}_hist"].sum()) == int(np.prod(z[f"{d}_shape"]))
    assert seen == set(range(8))
    assert z["trainA_hist"].tolist() == [347241, 14506, 1145, 0, 905627, 302871, 419595, 22543]
    assert z["trainB_hist"].tolist() == [315735, 42659, 3858, 2309, 860277, 611938, 38722, 221654]


def test_segclass_every_map_entry_and_default():
    cols = np.array([k for k, _ in O.CITYSCAPE_MAP] + [(0, 0, 0), (1, 2, 3), (128, 64, 129), (255, 255, 255)], np.uint8)
    exp = np.array([v for _, v in O.CITYSCAPE_MAP] + [0, 0, 0, 0], np.uint8)
    assert np.array_equal(O.seg_class_map(cols[None])[0], exp)
    assert O.seg_class_map(np.zeros((0, 5, 3), np.uint8)).shape == (0, 5)


def test_one_hot_and_resample():
    idx = np.array([[0, 3], [2, 1]])
    h = O.one_hot(idx, 4)
    assert h.shape == (2, 2, 4) and h.sum() == 4 and all(h[i, j, idx[i, j]] == 1 for i in range(2) for j in range(2))
    big = np.repeat(np.repeat(np.arange(16).reshape(4, 4), 32, 0), 32, 1)           # 128x128 blocks
    assert np.array_equal(O.resample_index_nearest(big, 4, 4), np.arange(16).reshape(4, 4))
    # literal reference call (cubic zoom of the one-hot) agrees with index-nearest on blocky maps
    ref = O.zoom_mask_reference(O.one_hot(big, 16), 128, 128)
    assert ref.shape == (4, 4, 16) and np.array_equal(ref, O.one_hot(O.resample_index_nearest(big, 4, 4), 16))


def test_oracle_small_regression():
    z = np.load(os.path.join(G, "oracle_small.npz"))
    PG = {k[3:]: z[k].astype(np.float64) for k in z.files if k.startswith("PG/")}
    PD = {k[3:]: z[k].astype(np.float64) for k in z.files if k.startswith("PD/")}
    real = (z["real_A_u8"].astype(np.float32) / np.float32(255)).astype(np.float64)
    seg = (z["seg_A_u8"].astype(np.float32) / np.float32(255)).astype(np.float64)
    mask = np.stack([O.one_hot(i, 34) for i in z["mask_idx"]]).astype(np.float64)
    r = O.train_step(PG, PD, real, seg, mask, n_blocks=2)
    assert abs(r["gen_loss"] - float(z["gen_loss"])) < 1e-10 and abs(r["disc_loss"] - float(z["disc_loss"])) < 1e-10
    assert np.abs(r["fake_A"] - z["fake_A"]).max() < 1e-6
    for k in PG:
        assert np.allclose(r["gG"][k], z["gG/" + k], rtol=1e-5, atol=1e-7 * max(1, np.abs(r["gG"][k]).max())), k
    for k in PD:
        assert np.allclose(r["PD"][k], z["newPD/" + k], rtol=1e-6, atol=1e-7), k


def test_oracle_full_fixture_is_wellformed():
    js = json.load(open(os.path.join(G, "oracle_full.json")))
    assert js["N"] == 2 and js["H"] == 128 and len(js["da_real"]) == 2 * 4 * 4
    assert set(js["gG_norm"]) == {n for n, _ in O.generator_param_shapes()}
    assert set(js["gD_norm"]) == {n for n, _ in O.discriminator_param_shapes()}
    # biases feeding an InstanceNorm get exactly zero gradient
    assert js["gG_norm"]["c2_b"] < 1e-9 and js["gD_norm"]["h1_b"] < 1e-9 and js["gG_norm"]["out_b"] > 0


def test_mask_rule_matches_the_reference_zoom_on_reference_data():
    """a12 pin: on three of the reference's own class-index maps (datasets/city/trainA_seg_class) the reference's call
    ``scipy.ndimage.zoom(one_hot(idx), (H/34/h, W/34/w, 1), mode="nearest")`` (utils.py:197-199; outputs committed by
    tests/golden/make_golden.py) equals one_hot of the align-corners nearest resample of the index map, at both loader grids."""
    z = np.load(os.path.join(G, "mask_zoom_city.npz"))
    cells = 0
    for i in range(len(z["names"])):
        idx = z[f"idx{i}"]
        assert idx.shape == (1024, 2048) and idx.max() < 34
        for H, W in z["sizes"]:
            exp = z[f"zoom{i}_{H}x{W}"]
            assert exp.shape == (round(H / 34), round(W / 34), 34) and (exp.sum(-1) == 1).all()
            got = O.mask_from_index(idx, 34, exp.shape[0], exp.shape[1])
            assert np.array_equal(got, exp.astype(np.float64))
            cells += exp.shape[0] * exp.shape[1]
    assert cells == 3 * (16 + 120)
