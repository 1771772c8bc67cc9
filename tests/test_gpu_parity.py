"""GPU parity: the HIP path (through the C ABI, libnasr.so) against the fp64 oracle on the same seeded
inputs.  Tolerances: logits <= 1e-4 max-abs, loss <= 1e-4 relative (we hold 2e-5),
every gradient tensor <= 1e-4 relative norm-wise (plus a small absolute floor for near-zero tensors),
post-Adam parameters <= 1e-4 relative, greedy decodes identical."""
import numpy as np
import pytest

from oracle import nasr_oracle as O

pytestmark = pytest.mark.gpu

MERGE = {'stack_reshape': 'stack_reshape', 'concat': 'concat', 'none': 'none'}


def make_engine(spec, lr=1e-3, graph=True):
    from neuralasr_amd.engine import Engine
    e = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, MERGE[spec.merge], spec.num_classes,
               forget_bias=spec.forget_bias, learning_rate=lr)
    e.set_graph_mode(graph)
    return e


def rand_params(spec, seed):
    rs = np.random.RandomState(seed)
    ps = O.init_params(spec, seed=seed)
    return [p + 0.05 * rs.randn(*p.shape) for p in ps]     # non-zero biases too


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


CASES = [
    # spec, B, T, var_len
    (O.ModelSpec(39, 24, 1, True, 'stack_reshape', 7), 4, 40, False),
    (O.ModelSpec(39, 24, 1, True, 'stack_reshape', 7), 4, 40, True),
    (O.ModelSpec(26, 40, 2, True, 'concat', 9), 6, 30, True),
    (O.ModelSpec(13, 32, 3, False, 'none', 5), 4, 25, True),
    (O.ModelSpec(20, 70, 1, True, 'concat', 29), 16, 20, True),
    (O.ModelSpec(20, 64, 1, True, 'stack_reshape', 29), 16, 33, True),
    (O.ModelSpec(10, 16, 1, False, 'none', 4), 1, 12, False),
    (O.ModelSpec(10, 16, 1, True, 'stack_reshape', 4), 1, 12, False),
    (O.ModelSpec(12, 20, 1, True, 'concat', 6), 20, 15, True),      # two M tiles
    (O.ModelSpec(12, 20, 2, False, 'none', 40), 33, 10, True),      # three M tiles, C > 32
    (O.ModelSpec(11, 32, 3, False, 'none', 6), 4, 75, True),        # a 3-layer unidirectional stack, T >= 64
    (O.ModelSpec(11, 24, 2, False, 'none', 6), 5, 133, True),       # 4 chunks, ragged last chunk
    # hidden sizes without a persistent instantiation (Hp > 512): the per-timestep kernels
    (O.ModelSpec(12, 600, 1, True, 'concat', 6), 40, 8, True),      # Hp 640: two-launch wide BPTT form, three M tiles
    (O.ModelSpec(12, 530, 1, False, 'none', 6), 3, 9, True),        # Hp 576 (not a multiple of 128): one-launch form, 18 partial sums
    # shape sweep: every padded hidden size class, M-tile count, class count and feature width the kernels branch on
    (O.ModelSpec(7, 150, 2, True, 'concat', 3), 7, 9, True),        # Hp 192
    (O.ModelSpec(30, 300, 1, True, 'stack_reshape', 29), 17, 8, True),   # Hp 320, two M tiles
    (O.ModelSpec(21, 400, 2, False, 'none', 12), 50, 7, True),      # Hp 448, four M tiles
    (O.ModelSpec(546, 500, 1, True, 'concat', 29), 64, 6, True),    # Hp 512 at the largest per-GPU batch
    (O.ModelSpec(5, 700, 1, True, 'concat', 4), 2, 7, False),       # Hp 704: per-step kernels, 22 partial sums
    (O.ModelSpec(9, 1000, 2, False, 'none', 8), 20, 6, True),       # Hp 1024: wide BPTT form, two M tiles, two layers
    (O.ModelSpec(1, 8, 1, True, 'concat', 3), 1, 5, False),         # one feature, two label classes + blank
    (O.ModelSpec(1100, 33, 1, False, 'none', 70), 9, 11, True),     # features wider than a GEMM K panel, C > 64
    (O.ModelSpec(14, 48, 4, True, 'concat', 150), 5, 14, True),     # four layers, C > 128
    # BASELINE.json configs[0] at its stated shape: lstm_ctc_net 1x128, 8 kHz / 13 MFCC (config/8000sr_13mfcc.config), batch 4
    (O.ModelSpec(13, 128, 1, False, 'none', 29), 4, 300, True),
    (O.ModelSpec(13, 128, 1, False, 'none', 29), 4, 300, False),
]


def case_id(c):
    s, B, T, v = c
    return f"F{s.feature_size}H{s.hidden}L{s.num_layers}{'bi' if s.bidirectional else 'uni'}-{s.merge}-C{s.num_classes}-B{B}T{T}{'v' if v else ''}"


@pytest.mark.parametrize("case", CASES, ids=case_id)
def test_forward_loss_grads_match_oracle(case):
    spec, B, T, var = case
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=100 + B + T, var_len=var,
                                                      Lmin=1, Lmax=max(1, T // 4))
    params = rand_params(spec, 3)
    e = make_engine(spec)
    assert e.backend == 'hip-gfx950'
    assert e.param_count == spec.param_count()
    e.set_params(O.flatten(params))
    np.testing.assert_array_equal(e.get_params(), O.flatten(params).astype(np.float32))

    loss_o, nll_o, grads_o, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    logits = e.forward(feats, seq_len)
    assert logits.shape == logits_o.shape
    # frames the CTC never reads (t >= seq_len) still hold defined values; compare everything
    np.testing.assert_allclose(logits, logits_o, atol=1e-4)

    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert loss == pytest.approx(loss_o, rel=2e-5)
    np.testing.assert_allclose(nll, nll_o, rtol=2e-5)
    gflat_o = O.flatten(grads_o)
    scale = np.linalg.norm(gflat_o)
    for (name, off, r, c), g_o in zip(e.tensors(), grads_o):
        g = grads[off:off + r * c].reshape(g_o.shape)
        err = np.linalg.norm(g - g_o)
        assert err <= 1e-4 * np.linalg.norm(g_o) + 1e-6 * scale, f'{name}: rel {rel(g, g_o):.3e}'
    assert rel(grads, gflat_o) < 1e-4

    # loss-only entry point agrees
    loss2, nll2 = e.loss(feats, seq_len, labels, label_len)
    assert loss2 == pytest.approx(loss, rel=1e-6)
    e.close()


@pytest.mark.parametrize("scaling", ['tiny', 'huge', 'mixed', 'sparse'])
def test_operand_scaling_of_the_fp16_plane_gemms(scaling):
    """The GEMMs split every operand into two fp16 planes under a power-of-two scale per operand row, measured on the
    device (gemm_tph.hip).  Inputs and weights far outside fp16's range, utterances 10 decades apart in one batch, and
    all-zero rows / columns must come out with the fp32 tolerances of the test above."""
    spec = O.ModelSpec(40, 96, 2, True, 'concat', 11)
    B, T = 6, 45
    F = spec.feature_size
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=77, var_len=True, Lmin=1, Lmax=9)
    params = rand_params(spec, 6)         # l0/fw/kernel [F+H, 4H], bias, l0/bw/kernel, bias, l1/..., W, b
    if scaling == 'tiny':          # features ~1e-7, first-layer input weights ~1e+5: products of order one
        feats = feats * 1e-7
        params[0][:F] *= 1e5
        params[2][:F] *= 1e5
    elif scaling == 'huge':        # features ~1e+6 (65504 is fp16's largest number), weights ~1e-7
        feats = feats * 1e6
        params[0][:F] *= 1e-7
        params[2][:F] *= 1e-7
    elif scaling == 'mixed':       # utterances 10 decades apart: one scale per frame row, one per feature column
        feats = feats * np.logspace(-5, 5, B)[:, None, None]
        params[0][:F] *= 1e-3
        params[2][:F] *= 1e-3
    else:                          # all-zero feature columns, an all-zero utterance, zero weight rows and columns
        feats[:, :, 5:17] = 0.0
        feats[2] = 0.0
        params[0][3:9] = 0.0
        params[0][:, 10:50] = 0.0
        params[4][:, :96] = 0.0
    e = make_engine(spec)
    e.set_params(O.flatten(params))
    pf = [np.asarray(p, np.float32).astype(np.float64) for p in params]          # what the engine holds
    ff = np.asarray(feats, np.float32).astype(np.float64)
    loss_o, nll_o, grads_o, logits_o = O.network_loss_and_grads(spec, pf, ff, seq_len, labels, label_len)
    logits = e.forward(ff.astype(np.float32), seq_len)
    np.testing.assert_allclose(logits, logits_o, atol=1e-4)
    loss, nll, grads = e.loss_and_grads(ff.astype(np.float32), seq_len, labels, label_len)
    assert np.isfinite(grads).all()
    assert loss == pytest.approx(loss_o, rel=2e-5)
    gscale = np.linalg.norm(O.flatten(grads_o))
    for (name, off, r, c), g_o in zip(e.tensors(), grads_o):
        g = grads[off:off + r * c].reshape(g_o.shape)
        assert np.linalg.norm(g - g_o) <= 1e-4 * np.linalg.norm(g_o) + 1e-6 * gscale, name
    e.close()


@pytest.mark.parametrize("case", CASES[:5], ids=case_id)
def test_greedy_decode_identical(case):
    spec, B, T, var = case
    feats, seq_len, _, _ = O.synth_batch(spec, B, T, seed=7, var_len=var)
    rs = np.random.RandomState(5)
    params = [p + 0.5 * rs.randn(*p.shape) for p in O.init_params(spec, seed=9)]
    e = make_engine(spec)
    e.set_params(O.flatten(params))
    logits_o, _ = O.network_forward(spec, [p.astype(np.float32).astype(np.float64) for p in params], feats, seq_len)
    want = O.greedy_decode(logits_o, seq_len)
    got = e.greedy_decode(feats, seq_len)
    # an argmax can only differ where the oracle's top-2 margin is below fp32 noise
    lg = e.forward(feats, seq_len)
    for b in range(B):
        if got[b] != want[b]:
            top2 = np.sort(logits_o[:seq_len[b], b], axis=-1)[:, -2:]
            assert (top2[:, 1] - top2[:, 0]).min() < 1e-4, f'utterance {b}: {got[b]} vs {want[b]}'
    assert O.greedy_decode(lg.astype(np.float64), seq_len) == got
    e.close()


@pytest.mark.parametrize("case", [CASES[1], CASES[2], CASES[3]], ids=case_id)
def test_train_steps_match_tf_adam(case):
    spec, B, T, var = case
    lr = 2e-3
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=21, var_len=var, Lmin=1, Lmax=max(1, T // 4))
    params = [p.astype(np.float32).astype(np.float64) for p in rand_params(spec, 4)]
    e = make_engine(spec, lr=lr)
    e.set_params(O.flatten(params))
    m = [np.zeros_like(p) for p in params]
    v = [np.zeros_like(p) for p in params]
    for step in range(1, 4):
        loss_o, _, g, _ = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
        params, m, v = O.adam_tf(params, g, m, v, step, lr)
        loss = e.train_step(feats, seq_len, labels, label_len)
        assert loss == pytest.approx(loss_o, rel=5e-5), f'step {step}'
    got = e.get_params()
    want = O.flatten(params)
    # Adam's first steps move every weight by ~lr regardless of gradient size: compare the UPDATE
    p0 = O.flatten([p.astype(np.float32) for p in rand_params(spec, 4)])
    assert rel(got - p0, want - p0) < 2e-3
    assert np.abs(got - want).max() < 2e-4     # <4% of the 3*lr a weight moves in three Adam steps
    gm, gv, gstep = e.get_adam_state()
    assert gstep == 3
    assert rel(gm, O.flatten(m)) < 1e-4
    assert rel(gv, O.flatten(v)) < 2e-4
    e.close()


def test_graph_and_eager_launch_paths_agree_bitwise():
    spec = O.ModelSpec(20, 48, 2, True, 'concat', 11)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 8, 21, seed=3, var_len=True, Lmin=1, Lmax=5)
    outs = []
    for graph in (True, False):
        e = make_engine(spec, graph=graph)
        e.set_params(O.flatten(rand_params(spec, 8)))
        loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
        outs.append((loss, grads))
        e.close()
    assert outs[0][0] == outs[1][0]
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_determinism_two_runs_bitwise():
    spec = O.ModelSpec(20, 64, 1, True, 'stack_reshape', 29)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 16, 40, seed=13, var_len=True)
    e = make_engine(spec)
    e.set_params(O.flatten(rand_params(spec, 2)))
    a = e.loss_and_grads(feats, seq_len, labels, label_len)
    b = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert a[0] == b[0]
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])     # every reduction of the step has a fixed order (no float atomics)
    e.close()


def test_infeasible_label_raises_value_error():
    from neuralasr_amd import _lib
    spec = O.ModelSpec(8, 16, 1, True, 'stack_reshape', 5)
    e = make_engine(spec)
    e.set_params(O.flatten(O.init_params(spec)))
    feats = np.zeros((2, 3, 8), np.float32)
    with pytest.raises(ValueError, match='Not enough time'):
        e.loss(feats, [2, 3], np.array([[1, 1, 0], [1, 2, 3]]), [2, 3])      # row 0: 2 labels + 1 repeat > 2 frames
    with pytest.raises(_lib.NasrError):
        e.loss(feats, [3, 3], np.array([[4, 0, 0], [1, 2, 3]]), [1, 3])      # id 4 == blank is not a label
    with pytest.raises(_lib.NasrError):
        e.loss(feats, [4, 3], np.array([[1, 0, 0], [1, 2, 3]]), [1, 3])      # seq_len > T
    e.close()
