"""GPU edge cases the reference's domain has: empty labels, single frames, the largest batch, long label
sequences (several lattice states per lane), a large symbol table (label_context >= 1 creates n-gram
symbols, preprocess_mfcc.py:22-26), hidden sizes that take every kernel template path."""
import numpy as np
import pytest

from oracle import nasr_oracle as O

pytestmark = pytest.mark.gpu


def engine_for(spec):
    from neuralasr_amd.engine import Engine
    return Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
                  learning_rate=1e-3)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def check(spec, feats, seq_len, labels, label_len, seed=3, gtol=1e-4):
    rs = np.random.RandomState(seed)
    params = [p + 0.05 * rs.randn(*p.shape) for p in O.init_params(spec, seed=seed)]
    params = [p.astype(np.float32).astype(np.float64) for p in params]
    e = engine_for(spec)
    e.set_params(O.flatten(params))
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    lo, nllo, go, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    np.testing.assert_allclose(e.forward(feats, seq_len), logits_o, atol=2e-4)
    assert loss == pytest.approx(lo, rel=3e-5)
    np.testing.assert_allclose(nll, nllo, rtol=3e-5, atol=1e-5)
    assert rel(grads, O.flatten(go)) < gtol
    e.close()


def test_empty_labels_and_single_frames():
    spec = O.ModelSpec(6, 16, 1, True, 'concat', 5)
    rs = np.random.RandomState(0)
    feats = rs.randn(5, 7, 6).astype(np.float32)
    seq_len = np.array([7, 1, 3, 1, 7], np.int32)
    for b in range(5):
        feats[b, seq_len[b]:] = 0
    labels = np.array([[1, 2, 3], [0, 0, 0], [0, 0, 0], [2, 0, 0], [1, 1, 2]], np.int32)
    label_len = np.array([3, 0, 0, 1, 3], np.int32)          # empty labels: nll = -sum log p(blank)
    check(spec, feats, seq_len, labels, label_len)


def test_all_labels_empty():
    spec = O.ModelSpec(6, 16, 1, False, 'none', 4)
    feats, seq_len, _, _ = O.synth_batch(spec, 3, 5, seed=1)
    check(spec, feats, seq_len, np.zeros((3, 1), np.int32), np.zeros(3, np.int32))


def test_T_equals_one():
    spec = O.ModelSpec(4, 16, 1, True, 'stack_reshape', 4)
    rs = np.random.RandomState(2)
    feats = rs.randn(3, 1, 4).astype(np.float32)
    check(spec, feats, np.array([1, 1, 1], np.int32), np.array([[1], [0], [2]], np.int32), np.array([1, 0, 1], np.int32))


@pytest.mark.parametrize("B", [17, 48, 64])
def test_batch_up_to_four_m_tiles(B):
    spec = O.ModelSpec(8, 24, 1, True, 'stack_reshape' if B == 64 else 'concat', 6)
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, 9, seed=B, var_len=True, Lmin=1, Lmax=3)
    check(spec, feats, seq_len, labels, label_len)


def test_batch_above_64_is_rejected():
    from neuralasr_amd import _lib
    spec = O.ModelSpec(8, 16, 1, True, 'concat', 6)
    e = engine_for(spec)
    feats = np.zeros((65, 4, 8), np.float32)
    with pytest.raises(_lib.NasrError, match=r'\[1,64\]'):
        e.forward(feats, [4] * 65)
    e.close()


@pytest.mark.parametrize("Lmax,T", [(40, 90), (100, 230), (200, 420)])
def test_long_labels_many_states_per_lane(Lmax, T):
    spec = O.ModelSpec(5, 16, 1, True, 'concat', 7)
    rs = np.random.RandomState(Lmax)
    B = 3
    seq_len = np.array([T, T - 7, T], np.int32)
    feats = rs.randn(B, T, 5).astype(np.float32)
    feats[1, T - 7:] = 0
    label_len = np.array([Lmax, Lmax // 2, 1], np.int32)
    labels = np.zeros((B, Lmax), np.int32)
    for b in range(B):
        # alternate labels so adjacent repeats stay rare and the label fits T
        labels[b, :label_len[b]] = (np.arange(label_len[b]) % 5) + rs.randint(0, 2, label_len[b]) * 0
    check(spec, feats, seq_len, labels, label_len, gtol=2e-4)


def test_label_too_long_for_lattice_kernel_is_an_error():
    from neuralasr_amd import _lib
    spec = O.ModelSpec(5, 16, 1, True, 'concat', 7)
    e = engine_for(spec)
    T, L = 1200, 520
    feats = np.zeros((1, T, 5), np.float32)
    labels = (np.arange(L) % 5).reshape(1, L).astype(np.int32)
    with pytest.raises(_lib.NasrError, match='label length'):
        e.loss(feats, [T], labels, [L])
    e.close()


def test_large_symbol_table():
    """label_context = 1 makes tri-gram symbols: C in the thousands."""
    spec = O.ModelSpec(7, 16, 1, True, 'concat', 3000)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 3, 12, seed=9, var_len=True, Lmin=2, Lmax=5)
    check(spec, feats, seq_len, labels, label_len)
    e = engine_for(spec)
    rs = np.random.RandomState(1)
    params = [p + 0.3 * rs.randn(*p.shape) for p in O.init_params(spec, seed=2)]
    e.set_params(O.flatten(params))
    lg = e.forward(feats, seq_len)
    assert e.greedy_decode(feats, seq_len) == O.greedy_decode(lg.astype(np.float64), seq_len)
    e.close()


@pytest.mark.parametrize("H,layers,bi", [(128, 1, True), (200, 2, True), (1024, 1, True), (1100, 1, False)])
def test_hidden_sizes_cover_every_template_path(H, layers, bi):
    """Hp = 128 / 256 / 1024 / 1152: NQ = 2, 4, 16 and the generic (runtime) forms; KSPLIT 4, 8, 32, 36."""
    spec = O.ModelSpec(9, H, layers, bi, 'concat' if bi else 'none', 6)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 4, 6, seed=H, var_len=True, Lmin=1, Lmax=2)
    check(spec, feats, seq_len, labels, label_len)


def test_device_side_context_stacking_is_bitwise_the_host_stacking():
    """include_context (utils.py:8-21) + the utterance-level normalisation (utils.py:29) done on the host, vs
    the same batch uploaded as its centre slice and stacked in HBM."""
    from neuralasr_amd import utils
    ctx, ncep, B, T = 3, 5, 4, 23
    spec = O.ModelSpec((2 * ctx + 1) * ncep, 24, 1, True, 'stack_reshape', 6)
    rs = np.random.RandomState(4)
    seq_len = np.array([23, 17, 9, 23], np.int32)
    feats = np.zeros((B, T, spec.feature_size), np.float32)
    for b in range(B):
        raw = rs.randn(seq_len[b], ncep).astype(np.float32)
        st = utils.include_context(raw, ctx, ncep)
        st = ((st - st.mean()) / st.std()).astype(np.float32)          # pad frames become (0-mean)/std
        feats[b, :seq_len[b]] = st
    _, _, labels, label_len = O.synth_batch(spec, B, 9, seed=2, Lmin=1, Lmax=3)
    e = engine_for(spec)
    e.set_params(O.flatten(O.init_params(spec, seed=5)))
    e.upload_batch(feats, seq_len, labels, label_len)
    e.compute_grads()
    l1, g1 = e.get_loss(), e.get_grads()
    assert e.upload_batch_context(feats, seq_len, labels, label_len, ctx, ncep) is True
    e.compute_grads()
    l2, g2 = e.get_loss(), e.get_grads()
    assert l1 == l2
    np.testing.assert_array_equal(g1, g2)
    # a batch without the window structure is refused (caller falls back to the plain upload)
    broken = feats.copy()
    broken[:, :, :ncep] = rs.randn(B, T, ncep)
    assert e.upload_batch_context(broken, seq_len, labels, label_len, ctx, ncep) is False
    e.close()


@pytest.mark.parametrize("spec", [O.ModelSpec(10, 32, 2, True, 'concat', 6), O.ModelSpec(10, 32, 3, False, 'none', 6)],
                         ids=['bi2', 'uni3'])
def test_shapes_change_between_steps(spec):
    """Real training feeds a different (B, T, Lmax) every step: buffers regrow, cached hipGraphs of a shape are reused
    when it returns, and nothing stale survives a re-allocation."""
    params = [p.astype(np.float32).astype(np.float64) for p in O.init_params(spec, seed=6)]
    e = engine_for(spec)
    e.set_params(O.flatten(params))
    for (B, T, seed) in [(4, 40, 1), (6, 96, 2), (4, 40, 1), (16, 70, 3), (2, 130, 4), (6, 96, 2), (17, 33, 5)]:
        feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=seed, var_len=True, Lmin=1, Lmax=max(2, T // 8))
        lo, _, go, _ = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
        loss, _, g = e.loss_and_grads(feats, seq_len, labels, label_len)
        assert loss == pytest.approx(lo, rel=3e-5), (B, T)
        assert rel(g, O.flatten(go)) < 1e-4, (B, T)
    e.close()


def test_staged_batches_equal_synchronous_uploads_bitwise():
    """nasr_stage_batch (pinned staging + copy stream, issued while a step is in flight) + nasr_commit_batch against
    nasr_upload_batch: same loss, same gradients, bit for bit - for plain features, for the context form, for the
    literal net's row map, with shapes changing from batch to batch; at most two batches ahead; discard returns a slot."""
    from neuralasr_amd import utils
    import threading
    ctx, ncep = 2, 4
    spec = O.ModelSpec((2 * ctx + 1) * ncep, 40, 1, True, 'stack_reshape', 7)
    rs = np.random.RandomState(11)

    def batch(B, T, seed):
        _, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=seed, var_len=True, Lmin=1, Lmax=3)
        feats = np.zeros((B, T, spec.feature_size), np.float32)
        for b in range(B):
            st = utils.include_context(rs.randn(int(seq_len[b]), ncep).astype(np.float32), ctx, ncep)
            feats[b, :seq_len[b]] = ((st - st.mean()) / st.std()).astype(np.float32)
        return feats, seq_len, labels, label_len
    batches = [batch(5, 19, 1), batch(3, 31, 2), batch(8, 12, 3), batch(5, 19, 4)]
    e, ref = engine_for(spec), engine_for(spec)
    p0 = O.flatten(O.init_params(spec, seed=5))
    e.set_params(p0)
    ref.set_params(p0)

    def ref_step(bt):
        ref.upload_batch(*bt)
        ref.compute_grads()
        out = ref.get_loss(), ref.get_grads()
        ref.apply_adam(1.0)
        return out
    want = [ref_step(bt) for bt in batches]
    # batch 0 the synchronous way; while its step is in flight, batches 1 (plain) and 2 (context form) are staged from
    # another thread; a third staged batch is refused; then they are committed in turn
    e.upload_batch(*batches[0])
    e.compute_grads()
    tickets = {}

    def loader():
        tickets[1] = e.stage_batch(*batches[1])
        tickets[2] = e.stage_batch(*batches[2], numcontext=ctx, numcep=ncep)
        tickets[3] = e.stage_batch(*batches[3])
    th = threading.Thread(target=loader)
    th.start()
    th.join()
    assert tickets[1] is not None and tickets[2] is not None and tickets[3] is None
    got = [(e.get_loss(), e.get_grads())]
    e.apply_adam(1.0)
    for k in (1, 2):
        e.commit_batch(tickets[k])
        if k == 2:
            tickets[3] = e.stage_batch(*batches[3], numcontext=ctx, numcep=ncep)     # a slot is free again
            assert tickets[3] is not None
        e.compute_grads()
        got.append((e.get_loss(), e.get_grads()))
        e.apply_adam(1.0)
    spare = e.stage_batch(*batches[0])
    e.discard_batch(spare)
    with pytest.raises(Exception, match='no staged batch'):
        e.commit_batch(spare)
    e.commit_batch(tickets[3])
    e.compute_grads()
    got.append((e.get_loss(), e.get_grads()))
    e.apply_adam(1.0)
    for (l1, g1), (l2, g2) in zip(got, want):
        assert l1 == l2
        np.testing.assert_array_equal(g1, g2)
    np.testing.assert_array_equal(e.get_params(), ref.get_params())
    # a validation batch in between (synchronous upload) always finds a slot, also with two batches staged
    t1, t2 = e.stage_batch(*batches[1]), e.stage_batch(*batches[2])
    assert t1 is not None and t2 is not None
    assert e.loss(*batches[0])[0] == ref.loss(*batches[0])[0]
    e.discard_batch(t1)
    e.discard_batch(t2)
    e.close()
    ref.close()


def test_per_step_graphs_follow_the_resident_batch():
    """The per-timestep fallback replays hipGraphs captured on the first batch of a shape: every later batch of that shape
    (other lengths, another upload slot) must be what the replay reads."""
    spec = O.ModelSpec(10, 40, 2, True, 'concat', 6)
    params = [p.astype(np.float32).astype(np.float64) for p in O.init_params(spec, seed=7)]
    e = engine_for(spec)
    e.set_params(O.flatten(params))
    e.set_recurrence_mode(False)
    assert e.recurrence_mode == 'per-step'
    for seed in (1, 2, 3, 4, 5):
        feats, seq_len, labels, label_len = O.synth_batch(spec, 6, 20, seed=seed, var_len=True, Lmin=1, Lmax=3)
        loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
        lo, nllo, go, _ = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
        assert loss == pytest.approx(lo, rel=3e-5), seed
        np.testing.assert_allclose(nll, nllo, rtol=3e-5)
        assert rel(grads, O.flatten(go)) < 1e-4
    e.close()


@pytest.mark.parametrize('spec,B,T', [
    (O.ModelSpec(13, 40, 2, True, 'concat', 7), 5, 37),           # B below the 16-row padding, odd T
    (O.ModelSpec(9, 24, 3, False, 'none', 6), 16, 50),            # one direction: only the frame-before lists
    (O.ModelSpec(20, 128, 1, True, 'stack_reshape', 9), 20, 64),  # the literal net
    (O.ModelSpec(20, 128, 2, True, 'concat', 9), 20, 64),         # persistent recurrence (Hp = 128), two batch blocks
           # persistent recurrence (Hp = 128), two batch blocks
])
def test_ragged_batches_are_computed_on_their_frames_only(spec, B, T):
    """DataSet.get_next_batch pads every utterance to the batch maximum (dataset.py:75-77).  With mostly-padding batches the
    operand passes and GEMMs cover the sum(seq_len) real frame rows only (nasr.h: nasr_set_row_compaction): the results must
    be the oracle's, and those of the same engine working on all T x B rows, whether the batch arrives by upload or through
    the staging slots; a full-length batch is compacted only where the batch dimension's own padding (to 16) is 15 % of
    the rows."""
    rs = np.random.RandomState(B + T)
    feats = rs.randn(B, T, spec.feature_size).astype(np.float32)
    seq_len = rs.randint(1, T // 2, size=B).astype(np.int32)
    seq_len[B // 2] = T                                     # one long utterance sets T; the others are mostly padding
    seq_len[0] = 1
    for b in range(B):
        feats[b, seq_len[b]:] = 0
    label_len = np.minimum(rs.randint(0, 6, size=B), seq_len // 2).astype(np.int32)
    labels = np.zeros((B, 6), np.int32)
    for b in range(B):
        labels[b, :label_len[b]] = rs.randint(0, spec.num_classes - 1, size=label_len[b])
    params = [p.astype(np.float32).astype(np.float64) for p in O.init_params(spec, seed=4)]
    lo, nllo, go, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    e = engine_for(spec)
    e.set_params(O.flatten(params))
    Bp = (B + 15) // 16 * 16
    res = {}
    for on in (True, False):
        e.set_row_compaction(on)
        loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
        assert e.resident_rows() == (int(seq_len.sum()) if on else T * Bp)
        assert loss == pytest.approx(lo, rel=3e-5)
        np.testing.assert_allclose(nll, nllo, rtol=3e-5, atol=1e-5)
        assert rel(grads, O.flatten(go)) < 1e-4
        res[on] = grads
    assert rel(res[True], res[False]) < 2e-6
    # through the staging slots, with another (full-length) batch resident in between
    e.set_row_compaction(True)
    full = np.full(B, T, np.int32)
    e.upload_batch(feats, full, labels, label_len)
    assert e.resident_rows() == (T * B if 20 * B <= 17 * Bp else T * Bp)
    ticket = e.stage_batch(feats, seq_len, labels, label_len)
    e.commit_batch(ticket)
    assert e.resident_rows() == int(seq_len.sum())
    e.compute_grads()
    np.testing.assert_array_equal(e.get_grads(), res[True])
    np.testing.assert_allclose(e.forward(feats, seq_len), logits_o, atol=2e-4)    # a forward-only batch: all rows
    assert e.resident_rows() == T * Bp
    e.close()


def test_ctc_lattice_on_flat_sharp_and_pinned_posteriors():
    """csrc/ctc.hip (2b), the default alpha / beta kernel (sorted base-2 three-term sums, emissions staged through LDS by a
    loader wave, columns rescaled every 4 frames): the oracle's loss and gradients (tf.nn.ctc_loss, networks/tfnetwork.py:58-59)
    on a fresh net, on the same net with its projection scaled by 60 (posteriors as sharp as a trained net's: most labels
    far below 1e-20 at most frames, a level that moves by 2^80 from one frame to the next) and on an utterance whose 35
    identical labels need 69 of its 70 frames (a handful of paths, every frame pinned)."""
    spec = O.ModelSpec(8, 16, 1, True, 'concat', 9)
    B, T = 6, 70
    rs = np.random.RandomState(11)
    feats = rs.randn(B, T, spec.feature_size).astype(np.float32)
    seq_len = np.array([70, 70, 64, 51, 70, 33], np.int32)
    for b in range(B):
        feats[b, seq_len[b]:] = 0
    label_len = np.array([12, 1, 20, 9, 30, 5], np.int32)
    labels = np.zeros((B, 35), np.int32)
    for b in range(B):
        labels[b, :label_len[b]] = rs.randint(0, spec.num_classes - 1, size=label_len[b])
    base = [p.astype(np.float32).astype(np.float64) for p in O.init_params(spec, seed=5)]
    e = engine_for(spec)
    names = [t[0] for t in e.tensors()]
    assert 'W' in names and 'b' in names
    pinned = labels.copy()
    pinned[0, :] = 3
    pinned_len = label_len.copy()
    pinned_len[0] = 35
    for scale, lab, ll in ((1.0, labels, label_len), (60.0, labels, label_len), (1.0, pinned, pinned_len)):
        params = [p * (scale if n in ('W', 'b') else 1.0) for n, p in zip(names, base)]
        e.set_params(O.flatten(params))
        loss, nll, grads = e.loss_and_grads(feats, seq_len, lab, ll)
        lo, nllo, go, _ = O.network_loss_and_grads(spec, params, feats, seq_len, lab, ll)
        assert loss == pytest.approx(lo, rel=3e-5)
        np.testing.assert_allclose(nll, nllo, rtol=3e-5, atol=1e-5)
        assert rel(grads, O.flatten(go)) < 1e-4
    e.close()
