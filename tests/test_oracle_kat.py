"""Hand-derivable known answers for the CPU oracle (SURVEY.md 8(c) list) and the
cross-check of the NumPy oracle against the independent torch-CPU composition."""
import numpy as np
import pytest

from oracle import sggan_oracle as O


def V(a):
    return O.Var(np.asarray(a, float))


def test_conv_same_s2_pads_after_only():
    # 4x4 ramp, all-ones 3x3 kernel, stride 2, SAME: TF pads (0,1) -> windows start at 0 and 2
    x = np.arange(16.0).reshape(1, 4, 4, 1)
    y = O.conv2d(O.Tape(), V(x), V(np.ones((3, 3, 1, 1))), V([0.0]), 2, "SAME").v[0, :, :, 0]
    xp = np.pad(x[0, :, :, 0], ((0, 1), (0, 1)))
    exp = np.array([[xp[0:3, 0:3].sum(), xp[0:3, 2:5].sum()], [xp[2:5, 0:3].sum(), xp[2:5, 2:5].sum()]])
    assert np.array_equal(y, exp)
    assert O.same_pads(4, 3, 2) == (0, 1, 2) and O.same_pads(5, 3, 2) == (1, 1, 3) and O.same_pads(7, 3, 1) == (1, 1, 7)


def test_deconv_impulse_placement_and_crop():
    # unit impulse at (i,j): out[2i+r, 2j+s] = w[r,s]; the LAST row/col of the full map is dropped
    w = np.arange(1.0, 10.0).reshape(3, 3, 1, 1)
    for (i, j) in ((0, 0), (1, 1)):
        x = np.zeros((1, 2, 2, 1)); x[0, i, j, 0] = 1
        y = O.deconv2d(O.Tape(), V(x), V(w), V([0.0])).v[0, :, :, 0]
        exp = np.zeros((5, 5)); exp[2 * i:2 * i + 3, 2 * j:2 * j + 3] = w[:, :, 0, 0]
        assert y.shape == (4, 4) and np.array_equal(y, exp[:4, :4])


def test_reflect_no_edge_repeat():
    assert O.reflect_index(5, 2).tolist() == [2, 1, 0, 1, 2, 3, 4, 3, 2]
    assert O.reflect_index(4, 3).tolist() == [3, 2, 1, 0, 1, 2, 3, 2, 1, 0]


def test_instance_norm_checkerboard():
    a, eps, g = 3.0, 1e-3, 1.7
    x = np.where((np.indices((4, 4)).sum(0) % 2) == 0, a, -a).reshape(1, 4, 4, 1)
    y = O.instance_norm(O.Tape(), V(x), V([g]), V([0.25]), eps).v
    assert np.allclose(np.abs(y - 0.25), g / np.sqrt(1 + eps / a ** 2), rtol=0, atol=1e-14)


def test_bce_l1_constants():
    z = V(np.zeros((2, 4, 4, 1)))
    assert abs(O.bce_logits_mean(O.Tape(), z, 1.0).v - np.log(2)) < 1e-15
    assert abs(O.bce_logits_mean(O.Tape(), z, 0.0).v - np.log(2)) < 1e-15
    assert O.l1_mean(O.Tape(), np.full((1, 2, 2, 3), 0.75), V(np.full((1, 2, 2, 3), 0.25))).v == 0.5


@pytest.mark.parametrize("eps,ratio", [(1e-7, None), (1e-2, None)])
def test_adam_tf_form(eps, ratio):
    lr, b1, b2 = 1e-3, 0.5, 0.999
    th, m, v = O.adam_tf(np.zeros(1), np.ones(1), np.zeros(1), np.zeros(1), 1, lr, b1, b2, eps)
    # TF form: -lr*sqrt(1-b2)/(sqrt(1-b2)+eps)   (torch form would be -lr/(1+eps))
    assert np.allclose(th, -lr * np.sqrt(1 - b2) / (np.sqrt(1 - b2) + eps), rtol=1e-14)
    if eps == 1e-2:
        assert abs(th[0] / -lr - 0.7597) < 1e-3 and abs(1 / (1 + eps) - 0.990) < 1e-3


def test_mask_reduce_is_channel_gather():
    rng = np.random.default_rng(0)
    h4 = rng.standard_normal((2, 1, 1, 34))
    idx = rng.integers(0, 34, (2, 4, 4))
    mask = np.stack([O.one_hot(i, 34) for i in idx]).astype(float)
    y = O.mask_reduce(O.Tape(), V(h4), mask).v
    assert y.shape == (2, 4, 4, 1)
    for n in range(2):
        assert np.array_equal(y[n, :, :, 0], h4[n, 0, 0][idx[n]])


def test_param_counts_match_survey():
    assert sum(int(np.prod(s)) for _, s in O.generator_param_shapes()) == 11_388_675
    assert sum(int(np.prod(s)) for _, s in O.discriminator_param_shapes()) == 8_791_970
    assert O.disc_out_hw(128, 128) == (1, 1) and O.disc_out_hw(256, 256) == (5, 5) and O.disc_out_hw(256, 512) == (5, 13)


def test_gradients_finite_difference():
    rng = np.random.default_rng(3)
    PG = O.init_params(O.generator_param_shapes(gf_dim=8, n_blocks=1), rng, 0.1)
    x = rng.uniform(0, 1, (1, 8, 8, 3)); tgt = rng.uniform(0, 1, (1, 8, 8, 3))

    def loss(P):
        t = O.Tape(); VP = {k: O.Var(v) for k, v in P.items()}
        out = O.generator_resnet(t, VP, O.Var(x), 1)
        l = O.l1_mean(t, tgt, out)
        return t, VP, l

    t, VP, l = loss(PG)
    t.backward([(l, 1.0)])
    for name in ("c1_w", "r1a_w", "r1b_g", "d1_w", "d2_b", "out_w", "c2_beta"):
        flat = PG[name].ravel(); i = int(rng.integers(flat.size)); h = 1e-6
        old = flat[i]
        flat[i] = old + h; lp = loss(PG)[2].v
        flat[i] = old - h; lm = loss(PG)[2].v
        flat[i] = old
        fd = (lp - lm) / (2 * h)
        assert abs(fd - VP[name].g.ravel()[i]) < 1e-6 * max(1, abs(fd)), name


def test_oracle_vs_torch_full_step_small_net():
    """NumPy oracle vs independent torch-CPU (float64) on the reduced networks, 2 steps."""
    import torch
    from oracle import torch_restatement as T
    rng = np.random.default_rng(7)
    PG = O.init_params(O.generator_param_shapes(gf_dim=8, n_blocks=2), rng, 0.1)
    PD = O.init_params(O.discriminator_param_shapes(df_dim=8), rng, 0.1)
    real = rng.uniform(0, 1, (2, 128, 128, 3)); seg = rng.uniform(0, 1, (2, 128, 128, 3))
    mask = np.stack([O.one_hot(i, 34) for i in rng.integers(0, 34, (2, 4, 4))]).astype(float)
    S = T.RefStep(PG, PD, torch.float64, n_blocks=2)
    st = None
    for t in (1, 2):
        r = O.train_step(PG, PD, real, seg, mask, st, t, n_blocks=2)
        o = S.step(real, seg, mask)
        assert abs(r["gen_loss"] - o["gen_loss"]) < 1e-10 and abs(r["disc_loss"] - o["disc_loss"]) < 1e-10
        assert np.abs(r["fake_A"] - o["fake_A"].numpy()).max() < 1e-10
        for k in PG:
            assert np.abs(r["gG"][k] - o["gG"][k].numpy()).max() < 1e-9 * max(1, np.abs(r["gG"][k]).max()), k
        for k in PD:
            assert np.abs(r["gD"][k] - o["gD"][k].numpy()).max() < 1e-9 * max(1, np.abs(r["gD"][k]).max()), k
        PG, PD, st = r["PG"], r["PD"], r["opt_state"]
        for k in PG:
            assert np.abs(PG[k] - S.PG[k].detach().numpy()).max() < 1e-8, k


def test_cycle_step_oracle_vs_torch():
    """NumPy cycle_step vs the independent torch composition (float64, reduced networks), both GAN criteria."""
    import torch
    from oracle import torch_restatement as T
    rng = np.random.default_rng(5)
    gs = O.generator_param_shapes(gf_dim=8, n_blocks=1); ds = O.discriminator_param_shapes(df_dim=8)
    P = {n: O.init_params(sh, rng, 0.1) for n, sh in (("Gab", gs), ("Gba", gs), ("Da", ds), ("Db", ds))}
    N, H, W = 1, 256, 256
    real_A, real_B = rng.uniform(0, 1, (N, H, W, 3)), rng.uniform(0, 1, (N, H, W, 3))
    pal = rng.integers(0, 256, (8, 3)) / 255.0
    seg_A = pal[np.repeat(np.repeat(rng.integers(0, 8, (N, 8, 8)), 32, 1), 32, 2)]
    seg_B = pal[np.repeat(np.repeat(rng.integers(0, 8, (N, 8, 8)), 32, 1), 32, 2)]
    mk = lambda: np.stack([O.one_hot(i, 34) for i in rng.integers(0, 34, (N, 5, 5))]).astype(float)
    mask_A, mask_B = mk(), mk()
    for lsgan in (True, False):
        r = O.cycle_step(P["Gab"], P["Gba"], P["Da"], P["Db"], real_A, real_B, seg_A, seg_B, mask_A, mask_B, use_lsgan=lsgan, n_blocks=1)
        o = T.CycleStep(P, torch.float64, use_lsgan=lsgan, n_blocks=1).step(real_A, real_B, seg_A, seg_B, mask_A, mask_B)
        assert abs(r["g_loss"] - o["g_loss"]) < 1e-10 and abs(r["d_loss"] - o["d_loss"]) < 1e-10
        assert np.abs(r["fake_B"] - o["fake_B"].numpy()).max() < 1e-10
        for n in r["grads"]:
            for k, g in r["grads"][n].items():
                assert np.abs(g - o["grads"][n][k].numpy()).max() < 1e-8 * max(1, np.abs(g).max()), (lsgan, n, k)
    # edge indicator: 1 exactly on the two columns next to a vertical class boundary
    seg = np.zeros((1, 5, 6, 3)); seg[:, :, 3:] = 0.5
    assert O.seg_edge_weight(seg)[0, :, :, 0].tolist() == [[0, 0, 1, 1, 0, 0]] * 5
