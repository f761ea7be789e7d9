"""Test infrastructure: read from a model's saved forward records (``sggan(keep_tapes=True).tapes``) which side of every ReLU /
LeakyReLU kink the HIP kernels took, in the order the float64 oracle evaluates its activations -- the ``branches`` argument of
``oracle.sggan_oracle.KinkPolicy``.  A layer's post-activation output is the next layer's saved input: it is positive exactly where
the kernel took the positive branch (ReLU maps the other side to 0, LeakyReLU keeps the sign)."""
import numpy as np


def _pos(t, c_real, sl=slice(None)):
    return (t[sl][..., :c_real] > 0).cpu().numpy()


def generator_branches(G, tape, sl=slice(None)):
    """Generator.forward record list -> one boolean array per ReLU, in forward order (c1, c2, c3, r1a ... rNa, d1, d2)."""
    nb = G.n_blocks
    out = [_pos(tape[1][1], G.c1.cout, sl), _pos(tape[2][1], G.c2.cout, sl), _pos(tape[3][0][1], G.c3.cout, sl)]
    for i in range(nb):
        out.append(_pos(tape[3 + i][1][1], G.blocks[i][0].cout, sl))          # r{i}a's output = r{i}b's input
    out.append(_pos(tape[4 + nb][1], G.d1.cout, sl))                          # d1's output = d2's input
    out.append(_pos(tape[5 + nb][1], G.d2.cout, sl))                          # d2's output = the head's input
    return out


def discriminator_branches(D, tape, sl=slice(None)):
    """Discriminator.forward record list -> one boolean array per LeakyReLU (h0 ... h33)."""
    return [_pos(tape[i + 1][1], u.cout, sl) for i, u in enumerate(D.units)]


def cycle_step_branches(m):
    """All activation decisions of one paired cycle step (``sggan(cycle=True, keep_tapes=True).tapes``) in the order
    ``oracle.cycle_step`` evaluates them: G_ab(real_A), G_ba(fake_B), G_ba(real_B), G_ab(fake_A), D_b(fake_B), D_a(fake_A),
    D_a(real_A), D_b(real_B).  The paired step stacks [first network's images; second network's images]:
    G_first = (G_ab(real_A); G_ba(real_B)), G_second = (G_ba(fake_B); G_ab(fake_A)).  The discriminators run either as two passes
    -- D_fake = (D_b(fake_B); D_a(fake_A)), D_real = (D_a(real_A); D_b(real_B)) -- or, with d_quad (the default, the path
    bench.py times), as ONE pass D_quad = (D_b(real_B); D_b(fake_B) | D_a(fake_A); D_a(real_A))."""
    t, n = m.tapes, m.tapes["n"]
    lo, hi = slice(0, n), slice(n, 2 * n)
    Gab, Gba, Da, Db = m.generator, m.generator_BA, m.discriminator, m.discriminator_B
    out = (generator_branches(Gab, t["G_first"], lo) + generator_branches(Gba, t["G_second"], lo)
           + generator_branches(Gba, t["G_first"], hi) + generator_branches(Gab, t["G_second"], hi))
    q = t.get("D_quad")
    if q is not None:
        out += (discriminator_branches(Db, q, slice(n, 2 * n)) + discriminator_branches(Da, q, slice(2 * n, 3 * n))
                + discriminator_branches(Da, q, slice(3 * n, 4 * n)) + discriminator_branches(Db, q, slice(0, n)))
    else:
        out += (discriminator_branches(Db, t["D_fake"], lo) + discriminator_branches(Da, t["D_fake"], hi)
                + discriminator_branches(Da, t["D_real"], lo) + discriminator_branches(Db, t["D_real"], hi))
    return out
