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
