"""HIP-graph replay of the train step (sggan_amd/graph.py): recorded steps must be the eager steps, bit for bit,
and the f32 golden-step parity must hold through the captured path."""
import numpy as np
import pytest
import torch

from tests.test_gpu_step import _rand_inputs, load_small, rel, small_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sg():
    import sggan_amd
    return sggan_amd


def _state(m):
    return [t.clone() for n in m.networks() for t in (n.P.flat, n.P.m, n.P.v, n.P.iterations, n.P.grad)] + [m._loss.clone()]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cycle", [False, True], ids=["reference", "cycle"])
def test_graph_replay_equals_eager_bitwise(sg, cycle, dtype):
    """3 steps with fresh inputs each step: parameters, Adam slots, step counters, gradients and losses after every step are
    bitwise those of the eager path (same kernels in the same order; Adam's t comes from the device counter)."""
    def run(graph):
        m = sg.sggan(sg.default_args(ngf=16, ndf=16, n_blocks=2, dtype=dtype, cycle=cycle, graph=graph))
        out = []
        for step in range(3):
            m.real_A, m.seg_A, m.mask_A = _rand_inputs(2, 256, 256, m.discriminator, 40 + step)
            if cycle:
                m.real_B, m.seg_B, m.mask_B = _rand_inputs(2, 256, 256, m.discriminator, 50 + step)
            m.train_step()
            out.append(_state(m) + [m.fake_A.tensor().clone()])
        return m, out
    me, eager = run(False)
    mg, graph = run(True)
    assert mg._program is not None and mg._program.n_graphs >= 1
    for step, (a, b) in enumerate(zip(eager, graph)):
        for i, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), (step, i)
    assert [n.P.step_count for n in mg.networks()] == [3] * len(mg.networks())


def test_golden_f32_step_through_the_captured_path(sg):
    """The f32 golden fixture (tests/golden/oracle_small.npz) checked on a step that ran as a HIP-graph replay."""
    m, z = small_model(sg, "f32")
    m.enable_graph()
    m.train_step()
    assert m._program is not None
    gl, dl = m.losses()
    assert abs(gl - float(z["gen_loss"])) < 1e-5 * abs(float(z["gen_loss"]))
    assert abs(dl - float(z["disc_loss"])) < 1e-5 * abs(float(z["disc_loss"]))
    assert rel(m.fake_A.numpy(), z["fake_A"]) < 1e-4
    for k, v in m.generator.P.export().items():
        if k.endswith("_b") and k != "out_b":
            continue
        assert np.abs(v - z["newPG/" + k]).max() < 2e-5, ("newPG", k)
    for k, v in m.discriminator.P.export().items():
        if k.endswith("_b") and k not in ("h0_b", "h4_b"):
            continue
        assert np.abs(v - z["newPD/" + k]).max() < 2e-5, ("newPD", k)


def test_graph_inputs_in_place_and_reshape(sg):
    """After the first graph step ``model.real_A`` etc. ARE the static buffers: filling them in place feeds the next
    replay without a copy; assigning inputs of another shape records a new program."""
    m = sg.sggan(sg.default_args(ngf=8, ndf=8, n_blocks=1, dtype="bf16", graph=True))
    e = sg.sggan(sg.default_args(ngf=8, ndf=8, n_blocks=1, dtype="bf16"))
    a = _rand_inputs(1, 256, 256, m.discriminator, 1)
    b = _rand_inputs(1, 256, 256, m.discriminator, 2)
    m.real_A, m.seg_A, m.mask_A = a
    m.train_step()
    prog = m._program
    bi = [m._convert_input(n, x) for n, x in zip(("real_A", "seg_A", "mask_A"), b)]
    m.real_A.copy_(bi[0]); m.seg_A.copy_(bi[1]); m.mask_A.copy_(bi[2])
    m.train_step()
    assert m._program is prog
    for inp in (a, b):
        e.real_A, e.seg_A, e.mask_A = inp
        e.train_step()
    assert torch.equal(m.generator.P.flat, e.generator.P.flat) and torch.equal(m._loss, e._loss)
    m.real_A, m.seg_A, m.mask_A = _rand_inputs(2, 256, 256, m.discriminator, 3)
    m.train_step()
    assert m._program is not prog
