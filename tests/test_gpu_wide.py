"""GPU: the wide persistent forward recurrence (neuralasr_amd/csrc/lstm_wide.hip: Hp = 2048 = the cell count of
networks/deepspeech.py:70-103, one launch per direction, the recurrent matrix resident in the registers of all 256 CUs)
against the per-timestep kernels of lstm.hip on the same inputs through the C ABI — every M-tile count, both stack
kinds, ragged lengths — and its abort / fall-back / re-arm path.  Parity of the same layer against the fp64 oracle at
DeepSpeech's own widths is tests/test_gpu_deepspeech.py::test_reference_widths_match_the_oracle."""
import os

import numpy as np
import pytest

from oracle import nasr_oracle as O

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('NASR_PERSIST', '1')[:1] == '0' or os.environ.get('NASR_WIDE', '1')[:1] == '0',
                                 reason='NASR_PERSIST=0 / NASR_WIDE=0 force the per-step kernels: nothing wide to test')]


def make_engine(spec, lr=1e-3):
    from neuralasr_amd.engine import Engine
    return Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
                  forget_bias=spec.forget_bias, learning_rate=lr)


def start_params(spec, seed):
    # float32 draws (the matrices hold 2 x 34 M weights): zero-mean, |U| ~ 0.03 keeps the recurrence well inside tanh's range
    rs = np.random.default_rng(seed)
    return rs.standard_normal(spec.param_count(), dtype=np.float32) * np.float32(0.03)


@pytest.mark.parametrize("bi,B,T", [(True, 3, 7), (True, 20, 6), (False, 40, 6), (True, 64, 5)],
                         ids=['bi-B3-1tile', 'bi-B20-2tiles', 'uni-B40-3tiles', 'bi-B64-4tiles'])
def test_wide_forward_equals_the_per_step_kernels(bi, B, T):
    spec = O.ModelSpec(10, 2048, 1, bi, 'concat' if bi else 'none', 5)
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=B + T, var_len=True, Lmin=1, Lmax=2)
    p0 = start_params(spec, 1)
    out = {}
    for mode in ('wide-persistent', 'per-step'):
        e = make_engine(spec)
        assert e.recurrence_mode == 'wide-persistent'
        if mode == 'per-step':
            e.set_recurrence_mode(False)
        assert e.recurrence_mode == mode
        e.set_params(p0)
        logits = e.forward(feats, seq_len)
        loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
        out[mode] = (logits, loss, grads)
        e.close()
    a, b = out['wide-persistent'], out['per-step']
    np.testing.assert_allclose(a[0], b[0], atol=2e-5)
    assert a[1] == pytest.approx(b[1], rel=2e-6)
    assert np.linalg.norm(a[2] - b[2]) <= 2e-5 * np.linalg.norm(b[2])


def test_aborted_wide_launch_voids_the_step_falls_back_and_is_rearmed(monkeypatch):
    """NASR_WIDE_FAULT=s: workgroup (0,0) treats the h hand-off of step s as timed out.  The launch drains through its
    bounded spins and raises the sticky error word and the fault word behind the gradients: that optimiser step is a
    no-op, the handle continues on the per-step kernels and returns to the wide kernel after NASR_PERSIST_REARM clean steps."""
    from neuralasr_amd import _lib
    spec = O.ModelSpec(10, 2048, 1, True, 'concat', 5)
    B, T = 5, 8
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=4, var_len=True, Lmin=1, Lmax=2)
    p0 = start_params(spec, 2)
    monkeypatch.setenv('NASR_PERSIST_REARM', '2')
    e, ref = make_engine(spec, lr=1e-4), make_engine(spec, lr=1e-4)
    ref.set_recurrence_mode(False)
    e.set_params(p0)
    ref.set_params(p0)
    assert e.recurrence_mode == 'wide-persistent' and e.persist_stats() == (0, 0)
    monkeypatch.setenv('NASR_WIDE_FAULT', '3')
    with pytest.raises(_lib.NasrError, match='persistent recurrence aborted'):
        e.train_step(feats, seq_len, labels, label_len)
    monkeypatch.delenv('NASR_WIDE_FAULT')
    assert e.recurrence_mode == 'per-step' and e.persist_stats() == (1, 0)
    np.testing.assert_array_equal(e.get_params(), p0)                 # the void step changed nothing
    assert e.get_adam_state()[2] == 0
    for want in ('per-step', 'per-step', 'wide-persistent', 'wide-persistent'):
        loss = e.train_step(feats, seq_len, labels, label_len)
        assert e.recurrence_mode == want
        assert loss == pytest.approx(ref.train_step(feats, seq_len, labels, label_len), rel=2e-5)
    assert e.persist_stats() == (1, 1)
    assert e.get_adam_state()[2] == ref.get_adam_state()[2] == 4
    assert np.abs(e.get_params() - ref.get_params()).max() < 0.05 * 4e-4
    e.close()
    ref.close()


@pytest.mark.parametrize("hook,value,code", [('NASR_WIDE_FAULT_BWD', '2', 'code 1'), ('NASR_WIDE_SCALE_SHIFT', '14', 'code 4')],
                         ids=['bptt-handoff-timeout', 'dG-beyond-the-fp16-planes'])
def test_aborted_wide_bptt_voids_the_step(monkeypatch, hook, value, code):
    """The BPTT kernel's two ways out: a hand-off that times out (NASR_WIDE_FAULT_BWD=s) and a dG * S_row that leaves the
    range of its fp16 planes (forced here by shifting the per-utterance scale, NASR_WIDE_SCALE_SHIFT).  Either way the
    step is void - parameters and Adam state untouched - and the same step on the per-step kernels matches an
    undisturbed engine."""
    from neuralasr_amd import _lib
    spec = O.ModelSpec(10, 2048, 1, True, 'concat', 5)
    B, T = 5, 8
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=4, var_len=True, Lmin=1, Lmax=2)
    p0 = start_params(spec, 2)
    e, ref = make_engine(spec, lr=1e-4), make_engine(spec, lr=1e-4)
    ref.set_recurrence_mode(False)
    e.set_params(p0)
    ref.set_params(p0)
    monkeypatch.setenv(hook, value)
    with pytest.raises(_lib.NasrError, match='persistent recurrence aborted \\(' + code):
        e.train_step(feats, seq_len, labels, label_len)
    monkeypatch.delenv(hook)
    assert e.recurrence_mode == 'per-step'
    np.testing.assert_array_equal(e.get_params(), p0)
    assert e.get_adam_state()[2] == 0
    assert e.train_step(feats, seq_len, labels, label_len) == pytest.approx(ref.train_step(feats, seq_len, labels, label_len), rel=2e-5)
    e.close()
    ref.close()


def test_full_length_pass_agrees_with_the_per_step_kernels():
    """BASELINE.json configs[3]'s recurrence at its own size (B 32, T 500, bidirectional, ragged lengths): 2 x 500 dependent
    timesteps through both wide kernels against 2 x 1000 per-step launches - loss, logits and every gradient tensor."""
    spec = O.ModelSpec(26, 2048, 1, True, 'concat', 29)
    B, T = 32, 500
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=11, var_len=True, Lmin=20, Lmax=60)
    p0 = start_params(spec, 5)
    out = {}
    for mode in ('wide-persistent', 'per-step'):
        e = make_engine(spec)
        if mode == 'per-step':
            e.set_recurrence_mode(False)
        assert e.recurrence_mode == mode
        e.set_params(p0)
        loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
        out[mode] = (loss, nll, grads, e.tensors())
        e.close()
    a, b = out['wide-persistent'], out['per-step']
    assert np.isfinite(a[2]).all() and a[0] == pytest.approx(b[0], rel=1e-6)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5)
    for name, off, r, c in a[3]:
        ga, gb = a[2][off:off + r * c], b[2][off:off + r * c]
        assert np.linalg.norm(ga - gb) <= 1e-4 * np.linalg.norm(gb) + 1e-7 * np.linalg.norm(b[2]), name
