"""GPU: the persistent recurrence (neuralasr_amd/csrc/lstm_persist.hip: one launch per layer pass, one XCD per
(direction, utterance slice)) against the fp64 oracle AND against the per-timestep kernels of lstm.hip on the same
inputs, through the C ABI.  Same tolerances as tests/test_gpu_parity.py.  The shapes exercise every
kernel instantiation family (Hp 64 / 128 / 192 / 256 / 320 / 512), several rounds over the batch (B > 16 bidirectional), partly filled
utterance slices, ragged lengths with T far above the shortest utterance, and the per-step fallback for hidden sizes the
persistent kernels do not cover."""
import numpy as np
import pytest

from oracle import nasr_oracle as O

import os

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('NASR_PERSIST', '1')[:1] == '0',
                                 reason='NASR_PERSIST=0 forces the per-step kernels: nothing persistent to test')]


def make_engine(spec, lr=1e-3, stream=None):
    from neuralasr_amd.engine import Engine
    return Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
                  forget_bias=spec.forget_bias, learning_rate=lr, stream=stream)


def rand_params(spec, seed):
    rs = np.random.RandomState(seed)
    return [p + 0.05 * rs.randn(*p.shape) for p in O.init_params(spec, seed=seed)]


CASES = [
    # spec, B, T                                                   Hp   what it exercises
    (O.ModelSpec(14, 50, 1, True, 'stack_reshape', 8), 5, 37),    # 64   NU 2, partly filled slices (B 5 of Bp 16)
    (O.ModelSpec(13, 128, 1, False, 'none', 6), 4, 60),           # 128  config 1 of BASELINE.json (lstm_ctc_net 1x128)
    (O.ModelSpec(20, 200, 2, True, 'concat', 11), 19, 24),        # 256  NU 8, 2 rounds (Bp 32), 2 layers
    (O.ModelSpec(26, 500, 1, True, 'stack_reshape', 29), 16, 48),  # 512  the literal BiLstmCTCNet width
    (O.ModelSpec(12, 460, 1, False, 'none', 7), 40, 18),          # 512 (padded from 460) uni, Bp 48: 2 rounds of 8 x 4
    (O.ModelSpec(12, 120, 3, False, 'none', 9), 3, 90),           # 128  3-layer uni stack, B 3
    (O.ModelSpec(15, 300, 1, True, 'concat', 8), 7, 22),          # 320  NU 10: partly filled second h load / output group
    (O.ModelSpec(11, 180, 2, False, 'none', 6), 10, 26),          # 192  NU 6
]


def case_id(c):
    s, B, T = c
    return f"H{s.hidden}L{s.num_layers}{'bi' if s.bidirectional else 'uni'}-B{B}T{T}"


@pytest.mark.parametrize("case", CASES, ids=case_id)
def test_persistent_matches_oracle_and_per_step_kernels(case):
    spec, B, T = case
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=7 * B + T, var_len=True, Lmin=1, Lmax=max(1, T // 5))
    seq_len[0] = max(int(label_len[0]) * 2 + 1, T // 3)            # one short utterance: many masked steps
    feats[0, seq_len[0]:] = 0.0
    params = rand_params(spec, 5)
    e = make_engine(spec)
    assert e.recurrence_mode == 'persistent', 'MI355X (8 XCDs x 32 CUs) must pass the census at create time'
    e.set_params(O.flatten(params))

    loss_o, nll_o, grads_o, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    logits_p = e.forward(feats, seq_len)
    np.testing.assert_allclose(logits_p, logits_o, atol=1e-4)
    loss_p, nll_p, grads_p = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert loss_p == pytest.approx(loss_o, rel=2e-5)
    np.testing.assert_allclose(nll_p, nll_o, rtol=2e-5)
    gflat_o = O.flatten(grads_o)
    scale = np.linalg.norm(gflat_o)
    for (name, off, r, c), g_o in zip(e.tensors(), grads_o):
        g = grads_p[off:off + r * c].reshape(g_o.shape)
        assert np.linalg.norm(g - g_o) <= 1e-4 * np.linalg.norm(g_o) + 1e-6 * scale, name

    # the same step through the per-timestep kernels: the two HIP paths agree far inside the oracle tolerance
    e.set_recurrence_mode(False)
    assert e.recurrence_mode == 'per-step'
    logits_s = e.forward(feats, seq_len)
    loss_s, nll_s, grads_s = e.loss_and_grads(feats, seq_len, labels, label_len)
    np.testing.assert_allclose(logits_p, logits_s, atol=2e-5)
    assert loss_p == pytest.approx(loss_s, rel=2e-6)
    assert np.linalg.norm(grads_p - grads_s) <= 2e-5 * np.linalg.norm(grads_s)
    # identical greedy decodes (north_star) in both modes
    hyp_s = e.greedy_decode(feats, seq_len)
    e.set_recurrence_mode(True)
    assert e.recurrence_mode == 'persistent'
    hyp_p = e.greedy_decode(feats, seq_len)
    assert hyp_p == hyp_s
    e.close()


def test_training_steps_agree_between_modes():
    """Three Adam steps in each mode from the same start: parameters stay together (the operand images of the
    persistent kernels are rebuilt after every update)."""
    spec = O.ModelSpec(16, 96, 2, True, 'concat', 10)
    B, T = 8, 30
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=11, var_len=True, Lmin=1, Lmax=6)
    start = O.flatten(rand_params(spec, 2)).astype(np.float32)
    out = []
    for persistent in (True, False):
        e = make_engine(spec, lr=1e-3)
        e.set_recurrence_mode(persistent)
        e.set_params(start)
        losses = [e.train_step(feats, seq_len, labels, label_len) for _ in range(3)]
        out.append((np.array(losses), e.get_params()))
        e.close()
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=5e-6)
    assert out[0][0][-1] < out[0][0][0]
    # Adam divides by sqrt(v): where a gradient is ~0 its fp32 rounding differences are amplified up to the step size
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=0, atol=5e-5)
    assert (np.abs(out[0][1] - out[1][1]) > 2e-6).mean() < 1e-3


def test_unsupported_width_uses_per_step_kernels():
    """Hp = 704 > 512 (22 units per CU: the slice of U no longer fits the registers) has no persistent instantiation: the engine reports and uses the per-step kernels,
    and asking for the persistent ones is an error, not a silent fallback."""
    spec = O.ModelSpec(10, 700, 1, True, 'concat', 5)
    B, T = 2, 9
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=3, Lmin=1, Lmax=3)
    params = rand_params(spec, 1)
    e = make_engine(spec)
    assert e.recurrence_mode == 'per-step'
    with pytest.raises(RuntimeError):
        e.set_recurrence_mode(True)
    e.set_params(O.flatten(params))
    loss_o, _, _, _ = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    loss, _, _ = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert loss == pytest.approx(loss_o, rel=2e-5)
    e.close()


def test_wide_layer_uses_the_two_launch_bptt_step():
    """Hp = 640 > 512: the BPTT step is lstm_bwd_cell_kernel (cell arithmetic once per cell) + the product kernel
    walking 4 K slices per block (lstm.hip); DeepSpeech's 2048-wide BiLSTM runs on this form."""
    spec = O.ModelSpec(9, 600, 1, True, 'concat', 6)
    B, T = 18, 7                                            # Bp = 32: two M tiles
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=21, var_len=True, Lmin=1, Lmax=2)
    params = rand_params(spec, 9)
    e = make_engine(spec)
    assert e.recurrence_mode == 'per-step'
    e.set_params(O.flatten(params))
    loss_o, nll_o, grads_o, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert loss == pytest.approx(loss_o, rel=2e-5)
    scale = np.linalg.norm(O.flatten(grads_o))
    for (name, off, r, c), g_o in zip(e.tensors(), grads_o):
        g = grads[off:off + r * c].reshape(g_o.shape)
        assert np.linalg.norm(g - g_o) <= 1e-4 * np.linalg.norm(g_o) + 1e-6 * scale, name
    e.close()


def test_aborted_persistent_launch_voids_the_step_and_falls_back(monkeypatch):
    """Fault injection (NASR_PERSIST_FAULT=s: one workgroup per group treats the hand-off of step s as timed out).
    The launch drains through its bounded spins, raises the sticky error word and the gradient buffer's fault word;
    Adam of that step is a no-op (parameters and Adam state untouched), the error surfaces at the next host sync, and
    the handle continues on the per-step kernels with correct results."""
    from neuralasr_amd import _lib
    spec = O.ModelSpec(12, 60, 2, True, 'concat', 7)
    B, T = 6, 16
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=5, var_len=True, Lmin=1, Lmax=3)
    params = rand_params(spec, 8)
    start = O.flatten(params).astype(np.float32)
    e = make_engine(spec, lr=1e-2)
    e.set_params(start)
    assert e.recurrence_mode == 'persistent'
    monkeypatch.setenv('NASR_PERSIST_FAULT', '3')
    with pytest.raises(_lib.NasrError, match='persistent recurrence aborted'):
        e.train_step(feats, seq_len, labels, label_len)
    monkeypatch.delenv('NASR_PERSIST_FAULT')
    assert e.recurrence_mode == 'per-step'
    np.testing.assert_array_equal(e.get_params(), start)              # the void step changed nothing
    m, v, step = e.get_adam_state()
    assert step == 0 and not m.any() and not v.any()
    with pytest.raises(_lib.NasrError):
        e.set_recurrence_mode(True)                                   # not offered again on this handle
    loss = e.train_step(feats, seq_len, labels, label_len)            # the same step on the per-step kernels
    loss_o, _, _, _ = O.network_loss_and_grads(spec, [p.astype(np.float32).astype(np.float64) for p in params], feats,
                                               seq_len, labels, label_len)
    assert loss == pytest.approx(loss_o, rel=2e-5)
    assert e.get_adam_state()[2] == 1 and np.abs(e.get_params() - start).max() > 1e-4
    e.close()


def test_persistent_mode_is_rearmed_after_clean_steps(monkeypatch):
    """After an abort the handle serves NASR_PERSIST_REARM clean steps on the per-step kernels, repeats the placement
    census and returns to the persistent kernels (include/nasr.h, nasr_get_persist_stats); a second abort doubles the
    wait.  The parameters follow the same trajectory as an undisturbed engine's throughout."""
    from neuralasr_amd import _lib
    spec = O.ModelSpec(12, 60, 2, True, 'concat', 7)
    B, T = 6, 16
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=5, var_len=True, Lmin=1, Lmax=3)
    start = O.flatten(rand_params(spec, 8)).astype(np.float32)
    monkeypatch.setenv('NASR_PERSIST_REARM', '2')
    e, ref = make_engine(spec, lr=1e-2), make_engine(spec, lr=1e-2)
    e.set_params(start)
    ref.set_params(start)
    assert e.recurrence_mode == 'persistent' and e.persist_stats() == (0, 0)

    def faulty_step():
        monkeypatch.setenv('NASR_PERSIST_FAULT', '3')
        with pytest.raises(_lib.NasrError, match='persistent recurrence aborted'):
            e.train_step(feats, seq_len, labels, label_len)
        monkeypatch.delenv('NASR_PERSIST_FAULT')

    def good_step(mode):
        loss = e.train_step(feats, seq_len, labels, label_len)
        assert e.recurrence_mode == mode
        assert loss == pytest.approx(ref.train_step(feats, seq_len, labels, label_len), rel=2e-5)

    faulty_step()
    assert e.recurrence_mode == 'per-step' and e.persist_stats() == (1, 0)
    good_step('per-step')
    good_step('per-step')
    good_step('persistent')                       # the third step starts with the census and re-arms
    assert e.persist_stats() == (1, 1)
    good_step('persistent')
    faulty_step()                                 # second abort: the wait doubles to 4 clean steps
    assert e.persist_stats() == (2, 1)
    for _ in range(4):
        good_step('per-step')
    good_step('persistent')
    assert e.persist_stats() == (2, 2)
    assert e.get_adam_state()[2] == ref.get_adam_state()[2] == 9
    np.testing.assert_allclose(e.get_params(), ref.get_params(), rtol=0, atol=2e-4)
    e.close()
    ref.close()


def test_fp16_plane_and_fp32_forward_recurrence_agree(monkeypatch):
    """The persistent forward recurrence keeps U as two fp16 planes and multiplies on v_mfma_f32_4x4x4_16B_f16 (three
    products per fp32 product, lstm_persist.hip); NASR_REC=f32 keeps fp32 planes and fp32 MFMAs.  Same logits, loss and
    gradients to fp32 rounding - on weights whose columns differ by 8 decades (one scale per gate column)."""
    spec = O.ModelSpec(15, 200, 2, True, 'concat', 8)
    B, T = 9, 40
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=13, var_len=True, Lmin=1, Lmax=7)
    params = rand_params(spec, 4)
    rs = np.random.RandomState(0)
    for i in (0, 2, 4, 6):                                  # kernels [I+H, 4H]: scale the recurrent rows' columns
        I = params[i].shape[0] - spec.hidden
        params[i][I:] *= 10.0 ** rs.uniform(-4, 0, size=(1, params[i].shape[1]))
    out = {}
    for mode in ('f16', 'f32'):
        monkeypatch.setenv('NASR_REC', mode)
        e = make_engine(spec)
        assert e.recurrence_mode == 'persistent'
        e.set_params(O.flatten(params))
        logits = e.forward(feats, seq_len)
        loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
        out[mode] = (logits, loss, grads)
        e.close()
    np.testing.assert_allclose(out['f16'][0], out['f32'][0], atol=2e-5)
    assert out['f16'][1] == pytest.approx(out['f32'][1], rel=2e-6)
    assert np.linalg.norm(out['f16'][2] - out['f32'][2]) <= 2e-5 * np.linalg.norm(out['f32'][2])
    loss_o, _, grads_o, logits_o = O.network_loss_and_grads(
        spec, [np.asarray(p, np.float32).astype(np.float64) for p in params], feats, seq_len, labels, label_len)
    np.testing.assert_allclose(out['f16'][0], logits_o, atol=1e-4)
    assert out['f16'][1] == pytest.approx(loss_o, rel=2e-5)



def test_no_abort_beside_a_cu_resident_collective_stand_in():
    """A kernel shaped like a ring all-reduce step (64 workgroups that hold their CUs while they sweep a 21 MB gradient
    bucket several times: nasr_diag_bucket_traffic) is released by the bucket events on a side stream in every step, with
    the events held back over the next persistent BPTT launch (the default).  50 steps: no persistent launch may give up -
    a collective beside the persistent kernels costs time, never the step - and the parameters stay those of an
    undisturbed engine, bit for bit (the stand-in leaves the gradients as they are)."""
    import torch
    spec = O.ModelSpec(546, 500, 3, True, 'concat', 29)
    B, T = 16, 120
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=3, var_len=True, Lmin=5, Lmax=20)
    p0 = O.flatten(O.init_params(spec, seed=1)).astype(np.float32)
    ts = torch.cuda.Stream()
    side = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        e = make_engine(spec, lr=1e-4, stream=ts.cuda_stream)
        ref = make_engine(spec, lr=1e-4, stream=ts.cuda_stream)
        for x in (e, ref):
            x.set_params(p0)
            assert x.recurrence_mode == 'persistent'
        e.set_bucket_defer(True)
        nb = len(e.grad_buckets())
        assert nb == 3
        for step in range(50):
            e.upload_batch(feats, seq_len, labels, label_len)
            e.compute_grads()
            for i in range(nb):
                e.diag_bucket_traffic(i, side.cuda_stream, 64, 6)
            ts.wait_stream(side)
            e.apply_adam(1.0)
            if step < 3:
                ref.train_step(feats, seq_len, labels, label_len)
            if step == 2:
                np.testing.assert_array_equal(e.get_params(), ref.get_params())
        torch.cuda.synchronize()
        assert not e.step_void()
        assert e.persist_stats() == (0, 0) and e.recurrence_mode == 'persistent'
        e.close()
        ref.close()


def test_side_stream_weight_gradients_are_bitwise_the_serial_ones():
    """with 500-wide layers and more than one layer, layer l's weight gradients run on a side stream beside the
    persistent BPTT launch of layer l-1, in a 3-wave GEMM instantiation with another tile shape - and with the K split of the
    serial order, so every gradient element sums the same k-blocks in the same order: loss and ALL gradients must be bit for
    bit those of the same engine with the overlap switched off, and so must the parameters after three optimiser steps."""
    spec = O.ModelSpec(546, 500, 3, True, 'concat', 29)
    B, T = 16, 150
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=9, var_len=True, Lmin=5, Lmax=25)
    p0 = O.flatten(O.init_params(spec, seed=1)).astype(np.float32)
    e = make_engine(spec, lr=1e-3)
    assert e.recurrence_mode == 'persistent'
    if not e.wgrad_overlap:
        e.close()
        pytest.skip('NASR_WGRAD_OVERLAP=0')
    out = {}
    for mode in (True, False, True):
        e.set_wgrad_overlap(mode)
        assert e.wgrad_overlap == mode
        e.set_params(p0)
        e.set_adam_state(np.zeros_like(p0), np.zeros_like(p0), 0)
        loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
        for _ in range(3):
            e.train_step(feats, seq_len, labels, label_len)
        res = (loss, nll, grads, e.get_params())
        if mode in out:
            prev = out[mode]
        else:
            prev = out.get(not mode)
            out[mode] = res
        if prev is not None:
            assert res[0] == prev[0]
            np.testing.assert_array_equal(res[1], prev[1])
            np.testing.assert_array_equal(res[2], prev[2])
            np.testing.assert_array_equal(res[3], prev[3])
    assert e.persist_stats() == (0, 0)
    e.close()
