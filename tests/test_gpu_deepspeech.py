"""GPU: the DeepSpeech-1 family (networks/deepspeech.py: clipped-ReLU dense stages with dropout around a BiLSTM,
BASELINE.json configs[3]) through the C ABI against the fp64 oracle.  The dropout keep-masks are a
hash of (seed, pass counter, stage, frame, utterance, unit) that the oracle computes identically (TensorFlow's random
stream cannot be reproduced: parity of the masks themselves is by definition, parity of everything else by comparison).
Tolerances as in tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from oracle import nasr_oracle as O

pytestmark = pytest.mark.gpu


def make_engine(spec, lr=1e-3):
    from neuralasr_amd.engine import Engine
    return Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
                  forget_bias=spec.forget_bias, learning_rate=lr, pre=spec.pre, post=spec.post, relu_clip=spec.relu_clip,
                  dropout=spec.dropout)


def rand_params(spec, seed, scale=0.15):
    rs = np.random.RandomState(seed)
    return [p + scale * rs.randn(*p.shape) for p in O.init_params(spec, seed=seed)]


CASES = [
    # spec, B, T
    (O.ModelSpec(20, 24, 1, True, 'concat', 7, pre=(40, 33, 50), post=36, relu_clip=1.0, dropout=(0.2, 0.1, 0.3, 0.25)), 5, 21),
    (O.ModelSpec(26, 70, 1, True, 'concat', 29, pre=(64, 64, 140), post=64, relu_clip=20.0, dropout=(0.05, 0.05, 0.05, 0.05)), 16, 30),
    (O.ModelSpec(12, 32, 1, True, 'concat', 6, pre=(48,), post=0, relu_clip=2.0, dropout=(0.5,)), 3, 17),      # one pre stage, no post stage
    (O.ModelSpec(12, 32, 2, False, 'none', 6, pre=(), post=20, relu_clip=0.7, dropout=(0.4,)), 4, 15),        # post stage only, uni stack
    (O.ModelSpec(16, 40, 1, True, 'concat', 9, pre=(32, 32, 80), post=32, relu_clip=3.0, dropout=()), 18, 12),  # no dropout
    # the reference's proportions at 1/8 width (n_hidden 256, cells 256, layer 3 = 512), 26 MFCC x 21 context, its own
    # dropout and clip: several GEMM tiles per stage, split-K weight gradients, the persistent recurrence at Hp 256
    (O.ModelSpec(546, 256, 1, True, 'concat', 29, pre=(256, 256, 512), post=256, relu_clip=20.0,
                 dropout=(0.05, 0.05, 0.05, 0.05)), 8, 60),
    # odd stage widths around a recurrence wider than the persistent kernels take (Hp 640: per-step kernels, wide BPTT
    # form), three M tiles
    (O.ModelSpec(20, 600, 1, True, 'concat', 7, pre=(70, 130), post=50, relu_clip=2.0, dropout=(0.1, 0.0, 0.2)), 33, 9),
]


def case_id(c):
    s, B, T = c
    return f"pre{'x'.join(map(str, s.pre)) or '0'}-H{s.hidden}L{s.num_layers}-post{s.post}-B{B}T{T}"


@pytest.mark.parametrize("case", CASES, ids=case_id)
def test_deepspeech_family_matches_oracle(case):
    spec, B, T = case
    params = rand_params(spec, 4)
    seed, counter = 4567, 11
    # the clipped ReLU is not differentiable at 0 and at the clip: a pre-activation within fp32 rounding (~1e-6 relative)
    # of a kink gets either mask in any fp32 implementation, and one flipped element moves a small gradient tensor by
    # 1e-3.  With 10^5..10^6 pre-activations some always lie within 1e-5: take, of 24 inputs, the one whose nearest
    # pre-activation is furthest from a kink, and require a clear margin.
    best = (-1.0, None)
    for data_seed in range(3 * B + T, 3 * B + T + 24):
        batch = O.synth_batch(spec, B, T, seed=data_seed, var_len=True, Lmin=1, Lmax=max(1, T // 5))
        margin = O.deepspeech_kink_margin(spec, params, batch[0], batch[1], drop=(seed, counter))
        if margin > best[0]:
            best = (margin, batch)
    assert best[0] > 1e-5, best[0]
    feats, seq_len, labels, label_len = best[1]
    e = make_engine(spec)
    names = [n for n, _, _, _ in e.tensors()]
    assert names == [n for n, _ in spec.param_shapes()]        # creation order of networks/deepspeech.py
    assert e.param_count == spec.param_count()
    e.set_params(O.flatten(params))
    np.testing.assert_array_equal(e.get_params(), O.flatten(params).astype(np.float32))

    loss_o, nll_o, grads_o, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len,
                                                                drop=(seed, counter))
    e.set_dropout_state(seed, counter)
    logits = e.forward(feats, seq_len)
    assert e.dropout_state() == (seed, counter + 1)            # every forward pass draws new masks
    np.testing.assert_allclose(logits, logits_o, atol=1e-4)
    e.set_dropout_state(seed, counter)
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert loss == pytest.approx(loss_o, rel=2e-5)
    np.testing.assert_allclose(nll, nll_o, rtol=2e-5)
    scale = np.linalg.norm(O.flatten(grads_o))
    for (name, off, r, c), g_o in zip(e.tensors(), grads_o):
        g = grads[off:off + r * c].reshape(g_o.shape)
        assert np.linalg.norm(g - g_o) <= 1e-4 * np.linalg.norm(g_o) + 1e-6 * scale, name
    if any(spec.drop_p(i) > 0 for i in range(4)):
        logits2 = e.forward(feats, seq_len)                    # counter moved on: other masks, other logits
        assert np.abs(logits2 - logits).max() > 1e-3
    e.close()


def test_deepspeech_training_step_matches_oracle_adam():
    spec = O.ModelSpec(14, 20, 1, True, 'concat', 6, pre=(24, 24, 40), post=24, relu_clip=2.0, dropout=(0.1, 0.1, 0.1, 0.1))
    B, T = 4, 14
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=8, var_len=True, Lmin=1, Lmax=3)
    params = rand_params(spec, 6)
    e = make_engine(spec, lr=1e-2)
    e.set_params(O.flatten(params))
    e.set_dropout_state(99, 0)
    st = O.AdamState(params) if hasattr(O, 'AdamState') else None
    losses = []
    p = [q.copy() for q in params]
    m = [np.zeros_like(q) for q in p]
    v = [np.zeros_like(q) for q in p]
    for step in range(3):
        losses.append(e.train_step(feats, seq_len, labels, label_len))
        loss_o, _, g, _ = O.network_loss_and_grads(spec, p, feats, seq_len, labels, label_len, drop=(99, step))
        assert losses[-1] == pytest.approx(loss_o, rel=5e-5)
        t = step + 1
        lr_t = 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        for k in range(len(p)):
            m[k] = 0.9 * m[k] + 0.1 * g[k]
            v[k] = 0.999 * v[k] + 0.001 * g[k] * g[k]
            p[k] = p[k] - lr_t * m[k] / (np.sqrt(v[k]) + 1e-8)
    got = e.get_params()
    np.testing.assert_allclose(got, O.flatten(p), rtol=0, atol=5e-4)
    e.close()


def test_reference_shape_constructs_and_steps():
    """networks/deepspeech.py at its own sizes (n_hidden 2048, BiLSTM 2048, 26-MFCC x 21 context), a short batch:
    the wide persistent kernels (lstm_wide.hip: one launch per direction and pass) serve Hp = 2048; loss finite and
    decreasing, and the same steps on the per-step kernels give the same losses and parameters."""
    from neuralasr_amd.networks.deepspeech import DeepSpeech
    spec = O.ModelSpec(546, DeepSpeech.n_cell_dim, 1, True, 'concat', 29, pre=DeepSpeech.pre_widths(), post=DeepSpeech.n_hidden,
                       relu_clip=DeepSpeech.relu_clip, dropout=DeepSpeech.dropout)
    B, T = 4, 24
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=1, Lmin=2, Lmax=4)
    rs = np.random.RandomState(0)
    runs = {}
    wide_on = os.environ.get('NASR_PERSIST', '1')[:1] != '0' and os.environ.get('NASR_WIDE', '1')[:1] != '0'
    for mode in (('wide-persistent', 'per-step') if wide_on else ('per-step',)):
        e = make_engine(spec, lr=1e-4)
        assert e.recurrence_mode == ('wide-persistent' if wide_on else 'per-step')
        if mode == 'per-step' and wide_on:
            e.set_recurrence_mode(False)
        assert e.recurrence_mode == mode
        if 'p0' not in runs:
            runs['p0'] = (rs.randn(e.param_count) * 0.02).astype(np.float32)
        e.set_params(runs['p0'])
        losses = [e.train_step(feats, seq_len, labels, label_len) for _ in range(3)]
        assert np.isfinite(losses).all() and losses[2] < losses[0]
        runs[mode] = (losses, e.get_params())
        e.close()
    if not wide_on:
        return
    np.testing.assert_allclose(runs['wide-persistent'][0], runs['per-step'][0], rtol=2e-5)
    d = runs['wide-persistent'][1] - runs['per-step'][1]
    # Adam's first steps move every weight by ~lr whatever its gradient: compare against that movement.  Element-wise the two
    # runs may part company where a last-bit difference puts a hidden unit on the other side of its ReLU / clip at one frame
    # (seen: 3 of the 2048 first-layer units, 62 weights, up to 1.2e-4) - bounded by the movement itself; in norm they agree.
    moved = runs['per-step'][1] - runs['p0']
    assert np.abs(d).max() <= 3 * 1e-4 * 1.01, np.abs(d).max()
    assert np.linalg.norm(d) < 1e-3 * np.linalg.norm(moved), np.linalg.norm(d) / np.linalg.norm(moved)
    assert (np.abs(d) > 0.05 * 3e-4).mean() < 1e-4


def _reference_spec():
    from neuralasr_amd.networks.deepspeech import DeepSpeech
    return O.ModelSpec(546, DeepSpeech.n_cell_dim, 1, True, 'concat', 29, pre=DeepSpeech.pre_widths(), post=DeepSpeech.n_hidden,
                       relu_clip=DeepSpeech.relu_clip, dropout=DeepSpeech.dropout)


def _params_away_from_kinks(spec, seed):
    """Weights of networks/deepspeech.py's shapes whose clipped-ReLU pre-activations sit at 4 +- ~0.5 (biases 4, weight
    scales set from the input moments): 10^7 pre-activations at the reference's widths, none within reach of a kink, so the
    fp32 path and the fp64 oracle take the same ReLU masks and can be compared to rounding.  (With natural weights a
    handful of the 10^7 land within fp32 rounding of 0 and each flipped mask element moves its layer's gradient by
    sqrt(1/N): see the second half of the test.)"""
    rs = np.random.RandomState(seed)
    params = []
    second_moment = {'h1': 1.0, 'h2': 17.0, 'h3': 17.0, 'h5': 0.1}
    for name, shp in spec.param_shapes():
        if name in second_moment:
            params.append(rs.randn(*shp) * 0.5 / np.sqrt(shp[0] * second_moment[name]))
        elif name in ('b1', 'b2', 'b3', 'b5'):
            params.append(np.full(shp, 4.0))
        elif name.endswith('kernel'):
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            k = rs.uniform(-lim, lim, size=shp)
            k[:shp[0] - spec.hidden] *= 0.15                     # the input rows see activations of magnitude 4
            params.append(k)
        elif name == 'h6':
            params.append(rs.randn(*shp) * np.sqrt(2.0 / (shp[0] + shp[1])))
        else:
            params.append(np.zeros(shp))
    return [p.astype(np.float32).astype(np.float64) for p in params]


def test_reference_widths_match_the_oracle():
    """BASELINE.json configs[3] per GPU: networks/deepspeech.py at its own widths (546 -> 2048 / 2048 / 4096 -> BiLSTM 2048
    -> 2048 -> 29), batch 32, dropout 0.05, against the fp64 oracle on a short ragged batch (T = 32: the oracle needs
    ~20 s here; the widths, not T, are what the smaller cases leave untested - 64-tile GEMMs, split-K weight gradients
    over 2048 / 4096-wide stages, the per-step recurrence kernels at Hp = 2048)."""
    spec = _reference_spec()
    B, T = 32, 32
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=5, var_len=True, Lmin=2, Lmax=6)
    seed, counter = 4567, 3
    params = _params_away_from_kinks(spec, 2)
    assert O.deepspeech_kink_margin(spec, params, feats, seq_len, drop=(seed, counter)) > 1e-2
    e = make_engine(spec)
    e.set_params(O.flatten(params))
    loss_o, nll_o, grads_o, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len,
                                                                drop=(seed, counter))
    e.set_dropout_state(seed, counter)
    logits = e.forward(feats, seq_len)
    np.testing.assert_allclose(logits, logits_o, atol=1e-4 * max(1.0, np.abs(logits_o).max()))
    e.set_dropout_state(seed, counter)
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert loss == pytest.approx(loss_o, rel=2e-5)
    np.testing.assert_allclose(nll, nll_o, rtol=2e-5)
    scale = np.linalg.norm(O.flatten(grads_o))
    worst = 0.0
    for (name, off, r, c), g_o in zip(e.tensors(), grads_o):
        g = grads[off:off + r * c].reshape(g_o.shape)
        err = np.linalg.norm(g - g_o) / (np.linalg.norm(g_o) + 1e-2 * scale / np.sqrt(len(grads_o)))
        worst = max(worst, err)
        assert err <= 1e-4, (name, err)
    # the same net with the reference's own initialisation (zero biases: pre-activations straddle the ReLU kink): the
    # forward pass is continuous across it - logits and loss to the usual tolerance; a gradient tensor may differ by the
    # few mask elements (of 10^7) that fp32 rounding decides the other way
    params = [p.astype(np.float32).astype(np.float64) for p in O.init_params(spec, seed=3)]
    e.set_params(O.flatten(params))
    loss_o, nll_o, grads_o, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len,
                                                                drop=(seed, counter))
    e.set_dropout_state(seed, counter)
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert loss == pytest.approx(loss_o, rel=5e-5)
    g_o = O.flatten(grads_o)
    assert np.linalg.norm(grads - g_o) <= 1e-2 * np.linalg.norm(g_o)
    e.close()


_LONG = {}


def _long_case():
    """configs[3] per GPU over > 100 dependent timesteps: B 32, ragged T <= 128, dropout 0.05, checked by oracle/cref (the C
    restatement, pinned against the fp64 oracle in tests/test_cref.py; the fp64 oracle would need ~10 minutes here).
    Computed once for both recurrence modes."""
    if not _LONG:
        from oracle import cref
        cref.set_threads(cref.usable_cpus())
        spec = _reference_spec()
        B, T = 32, 128
        batch = O.synth_batch(spec, B, T, seed=7, var_len=True, Lmin=10, Lmax=30)
        seed, counter = 4567, 3
        params = _params_away_from_kinks(spec, 2)
        assert O.deepspeech_kink_margin(spec, params, batch[0], batch[1], drop=(seed, counter)) > 1e-2
        flat = O.flatten(params).astype(np.float32)
        lo, nllo, go, lgo = cref.loss_and_grads(spec, flat, *batch, want_logits=True, drop=(seed, counter))
        _LONG.update(spec=spec, batch=batch, drop=(seed, counter), flat=flat, ref=(lo, nllo, go, lgo))
    return _LONG


@pytest.mark.parametrize("mode", ['wide-persistent', 'per-step'])
def test_reference_widths_over_128_timesteps_match_the_c_restatement(mode):
    """networks/deepspeech.py:70-121 at its own widths over 64..128 dependent timesteps per utterance: logits, loss and
    every gradient tensor against oracle/cref, once through the wide persistent kernels (fp16-plane BPTT with its
    per-utterance dG scale, lstm_wide.hip) and once through the per-step kernels at Hp = 2048 (lstm.hip) - each pinned to a
    CPU restatement instead of to the other."""
    if mode == 'wide-persistent' and (os.environ.get('NASR_PERSIST', '1')[:1] == '0' or os.environ.get('NASR_WIDE', '1')[:1] == '0'):
        pytest.skip('NASR_PERSIST=0 / NASR_WIDE=0 force the per-step kernels')
    c = _long_case()
    spec, (feats, seq_len, labels, label_len), (seed, counter) = c['spec'], c['batch'], c['drop']
    lo, nllo, go, lgo = c['ref']
    e = make_engine(spec)
    if mode == 'per-step':
        e.set_recurrence_mode(False)
    assert e.recurrence_mode == mode
    e.set_params(c['flat'])
    e.set_dropout_state(seed, counter)
    logits = e.forward(feats, seq_len)
    np.testing.assert_allclose(logits, lgo, atol=1e-4 * max(1.0, np.abs(lgo).max()))
    e.set_dropout_state(seed, counter)
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert e.recurrence_mode == mode                       # no abort, no fall-back on the way
    assert loss == pytest.approx(lo, rel=2e-5)
    np.testing.assert_allclose(nll, nllo, rtol=2e-5)
    scale = np.linalg.norm(go)
    for name, off, r, cc in e.tensors():
        g, g_o = grads[off:off + r * cc], go[off:off + r * cc]
        err = np.linalg.norm(g - g_o) / (np.linalg.norm(g_o) + 1e-2 * scale / np.sqrt(len(e.tensors())))
        assert err <= 1e-4, (name, err)
    assert np.linalg.norm(grads - go) <= 1e-4 * scale
    e.close()


def test_lstm_gradients_form_a_bucket_of_their_own():
    """include/nasr.h "Overlapping the exchange": a DeepSpeech-shaped net finishes the (Bi)LSTM's gradients + W + b before
    the backward pass of the dense stages in front of it, so they are bucket 0 (released after the layer's weight
    gradients) and the dense stages + the fault word the last bucket; the two tile the device buffer."""
    spec = CASES[1][0]
    e = make_engine(spec)
    buckets = e.grad_buckets()
    assert len(buckets) == 2 and buckets[-1][0] == 0 and buckets[0][0] == buckets[-1][1]
    assert sum(c for _, c in buckets) == e.grad_device_ptr()[1]
    H, D = spec.hidden, 2
    assert buckets[0][1] >= (spec.pre[-1] + H) * 4 * H * D          # at least the layer's kernels
    e.close()
