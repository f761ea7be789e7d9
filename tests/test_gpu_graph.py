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


def test_eager_inference_between_graph_replays_sees_the_updated_weights(sg):
    """A replayed step runs Adam on the device; the host-side re-pack trigger (ParamStore.version) must move with it, or an
    eager ``generator(x)`` after a replay (test_during_train at every epoch end, model.py:263) would convolve with packed
    weights that are one or more steps old.  graph steps -> eager generator -> graph steps -> eager generator, against an
    all-eager model: bitwise the same images."""
    def run(graph):
        m = sg.sggan(sg.default_args(ngf=8, ndf=8, n_blocks=1, dtype="bf16", graph=graph))
        x = torch.rand((1, 256, 256, 3), generator=torch.Generator().manual_seed(9)).cuda()
        outs = [m.generator(x).clone()]
        for rnd in range(3):
            for step in range(2):
                m.real_A, m.seg_A, m.mask_A = _rand_inputs(1, 256, 256, m.discriminator, 70 + 2 * rnd + step)
                m.train_step()
            outs.append(m.generator(x).clone())
        m.enable_graph(False)                              # ... and an eager train step after replays
        m.real_A, m.seg_A, m.mask_A = _rand_inputs(1, 256, 256, m.discriminator, 99)
        m.train_step()
        outs.append(m.generator(x).clone())
        return outs
    eager, graph = run(False), run(True)
    assert not torch.equal(eager[0], eager[1]) and not torch.equal(eager[1], eager[2])      # the weights do move
    for i, (a, b) in enumerate(zip(eager, graph)):
        assert torch.equal(a, b), i


def test_recorded_program_keeps_its_workspace_alive(sg):
    """The scratch workspace is allocated outside the capture pool and its address is baked into the recorded launches: a
    later, larger request must not free it under the program (it would be handed to another tensor and silently overwritten
    by every replay).  Record a step, force the workspace to grow, allocate over the freed pool, replay: same result as a
    model that never saw the growth."""
    from sggan_amd import kernels as K
    def run(disturb):
        m = sg.sggan(sg.default_args(ngf=8, ndf=8, n_blocks=1, dtype="f32", graph=True, cycle=True))
        m.real_A, m.seg_A, m.mask_A = _rand_inputs(1, 256, 256, m.discriminator, 5)
        m.real_B, m.seg_B, m.mask_B = _rand_inputs(1, 256, 256, m.discriminator, 6)
        m.train_step()
        if disturb:
            old = K.workspace(1, m.device)
            assert any(t is old for t in m._program.keep)
            big = K.workspace(old.numel() * 2 + (64 << 20), m.device)          # replaces the shared buffer
            assert big is not old
            del old
            junk = [torch.full((1 << 20,), float("nan"), device=m.device) for _ in range(64)]   # would land in the freed block
        m.train_step()
        return [n.P.flat.clone() for n in m.networks()] + [m._loss.clone()]
    for a, b in zip(run(False), run(True)):
        assert torch.equal(a, b)
