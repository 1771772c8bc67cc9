"""Pins for the CPU oracle (oracle/nasr_oracle.py).  The reference ships no tests or golden
vectors for this path, so the oracle is pinned by: brute-force CTC enumeration,
finite differences, and an independent torch-CPU implementation (autograd LSTM with TF gate
order + F.ctc_loss(blank=C-1))."""
import numpy as np
import pytest
import torch

from oracle import nasr_oracle as O


# ----------------------------------------------------------------- CTC known answers
@pytest.mark.parametrize("T,C,label", [
    (1, 3, []), (3, 3, []), (2, 3, [0]), (4, 3, [0, 1]), (5, 4, [1, 1]), (6, 4, [0, 1, 0]),
    (6, 3, [0, 0, 1]), (5, 4, [2, 2, 2]), (7, 3, [1, 0, 1]),
])
def test_ctc_matches_alignment_enumeration(T, C, label):
    rs = np.random.RandomState(T * 31 + C)
    logits = rs.randn(T, C) * 2
    nll, grad, _, _ = O.ctc_single(logits, label, C - 1)
    assert nll == pytest.approx(O.ctc_brute_force(logits, label, C - 1), rel=1e-12, abs=1e-12)
    # gradient of the enumerated loss by central differences
    eps = 1e-6
    for t in range(T):
        for k in range(C):
            lp, lm = logits.copy(), logits.copy()
            lp[t, k] += eps
            lm[t, k] -= eps
            fd = (O.ctc_brute_force(lp, label, C - 1) - O.ctc_brute_force(lm, label, C - 1)) / (2 * eps)
            assert grad[t, k] == pytest.approx(fd, abs=2e-8)


def test_ctc_infeasible_raises_like_tf():
    with pytest.raises(ValueError, match="Not enough time"):
        O.ctc_single(np.zeros((2, 4)), [1, 1], 3)     # 2 labels + 1 separating blank > 2 frames
    with pytest.raises(ValueError):
        O.ctc_single(np.zeros((2, 4)), [0, 1, 2], 3)


def test_ctc_matches_torch():
    rs = np.random.RandomState(0)
    Tp, B, C, Lmax = 23, 5, 7, 6
    logits = rs.randn(Tp, B, C)
    seq_len = np.array([23, 20, 17, 23, 9])
    label_len = np.array([6, 3, 0, 5, 4])
    labels = rs.randint(0, C - 1, size=(B, Lmax))
    labels[3, :5] = [2, 2, 3, 3, 2]
    nll, grad = O.ctc_loss_and_grad(logits, labels, label_len, seq_len)
    lt = torch.tensor(logits, requires_grad=True)
    tl = torch.nn.functional.ctc_loss(torch.log_softmax(lt, -1), torch.tensor(labels), torch.tensor(seq_len),
                                      torch.tensor(label_len), blank=C - 1, reduction='none', zero_infinity=False)
    tl.sum().backward()
    np.testing.assert_allclose(nll, tl.detach().numpy(), rtol=1e-10)
    np.testing.assert_allclose(grad, lt.grad.numpy(), atol=1e-10)


# ----------------------------------------------------------------- LSTM single step by hand
def test_lstm_single_step_hand_computed():
    # I=1, H=1, one frame: g = x*k_x + 0*k_h + bias; i,j,f,o order
    K = np.array([[0.5, -1.0, 2.0, 0.25], [9., 9., 9., 9.]])
    b = np.array([0.1, 0.2, -0.3, 0.4])
    x = np.array([[[2.0]]])
    out, _ = O.lstm_dir_forward(x, [1], K, b, forget_bias=1.0)
    g = 2.0 * K[0] + b
    sig = lambda v: 1 / (1 + np.exp(-v))
    c = 0 * sig(g[2] + 1.0) + sig(g[0]) * np.tanh(g[1])
    assert out[0, 0, 0] == pytest.approx(np.tanh(c) * sig(g[3]), rel=1e-14)


# ----------------------------------------------------------------- torch re-implementation
def _torch_lstm_dir(x, seq_len, K, b, fb, reverse):
    B, T, I = x.shape
    H = K.shape[1] // 4
    h = torch.zeros(B, H, dtype=x.dtype)
    c = torch.zeros(B, H, dtype=x.dtype)
    outs = [[None] * T for _ in range(B)]
    zero = torch.zeros(H, dtype=x.dtype)
    for s in range(T):
        rows = []
        for bb in range(B):
            if s < seq_len[bb]:
                rows.append(x[bb, seq_len[bb] - 1 - s if reverse else s])
            else:
                rows.append(torch.zeros(I, dtype=x.dtype))
        xt = torch.stack(rows)
        g = torch.cat([xt, h], 1) @ K + b
        i, j, f, o = g.split(H, 1)
        cn = c * torch.sigmoid(f + fb) + torch.sigmoid(i) * torch.tanh(j)
        hn = torch.tanh(cn) * torch.sigmoid(o)
        mask = torch.tensor([[1.0 if s < seq_len[bb] else 0.0] for bb in range(B)], dtype=x.dtype)
        c = mask * cn + (1 - mask) * c
        h = mask * hn + (1 - mask) * h
        for bb in range(B):
            if s < seq_len[bb]:
                outs[bb][seq_len[bb] - 1 - s if reverse else s] = hn[bb]
    return torch.stack([torch.stack([o if o is not None else zero for o in row]) for row in outs])


def _torch_net(spec, params, feats, seq_len, labels, label_len):
    x = torch.tensor(feats, dtype=torch.float64)
    B, T, _ = x.shape
    H, C = spec.hidden, spec.num_classes
    pi = 0
    for l in range(spec.num_layers):
        if spec.bidirectional:
            of = _torch_lstm_dir(x, seq_len, params[pi], params[pi + 1], spec.forget_bias, False)
            ob = _torch_lstm_dir(x, seq_len, params[pi + 2], params[pi + 3], spec.forget_bias, True)
            pi += 4
            last = (of, ob)
            x = torch.cat([of, ob], 2)
        else:
            x = _torch_lstm_dir(x, seq_len, params[pi], params[pi + 1], spec.forget_bias, False)
            pi += 2
            last = (x,)
    if spec.bidirectional and spec.merge == 'stack_reshape':
        flat = torch.stack(last, 0).reshape(-1, H)
    else:
        flat = x.reshape(-1, spec.proj_in)
    logits = (flat @ params[-2] + params[-1]).reshape(B, -1, C).permute(1, 0, 2)
    nll = torch.nn.functional.ctc_loss(torch.log_softmax(logits, -1), torch.tensor(labels),
                                       torch.tensor(seq_len), torch.tensor(label_len), blank=C - 1, reduction='none')
    return logits, nll.mean()


SPECS = [
    O.ModelSpec(5, 4, 1, True, 'stack_reshape', 5),
    O.ModelSpec(5, 4, 2, True, 'concat', 6),
    O.ModelSpec(3, 5, 3, False, 'none', 4),
    O.ModelSpec(4, 3, 1, True, 'concat', 5),
]


@pytest.mark.parametrize("spec", SPECS, ids=lambda s: f"L{s.num_layers}{'bi' if s.bidirectional else 'uni'}-{s.merge}")
def test_network_matches_torch_autograd(spec):
    rs = np.random.RandomState(7)
    B, T = 4, 9
    seq_len = np.array([9, 7, 9, 5])
    feats = rs.randn(B, T, spec.feature_size)
    for b in range(B):
        feats[b, seq_len[b]:] = 0
    label_len = np.array([3, 2, 1, 2])
    labels = rs.randint(0, spec.num_classes - 1, size=(B, 3))
    params = [p + 0.1 * rs.randn(*p.shape) for p in O.init_params(spec, seed=3)]
    loss, nll, grads, logits = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    tp = [torch.tensor(p, requires_grad=True) for p in params]
    tlogits, tloss = _torch_net(spec, tp, feats, seq_len, labels, label_len)
    tloss.backward()
    np.testing.assert_allclose(logits, tlogits.detach().numpy(), atol=1e-12)
    assert loss == pytest.approx(float(tloss), rel=1e-12)
    for (name, _), g, t in zip(spec.param_shapes(), grads, tp):
        np.testing.assert_allclose(g, t.grad.numpy(), atol=1e-11, err_msg=name)


def test_network_finite_differences():
    spec = O.ModelSpec(3, 3, 1, True, 'stack_reshape', 4)
    rs = np.random.RandomState(11)
    B, T = 2, 6
    seq_len = np.array([6, 4])
    feats = rs.randn(B, T, 3)
    feats[1, 4:] = 0
    labels = np.array([[1, 2], [0, 0]])
    label_len = np.array([2, 1])
    params = [p + 0.2 * rs.randn(*p.shape) for p in O.init_params(spec, seed=5)]
    _, _, grads, _ = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    flat = O.flatten(params)
    gflat = O.flatten(grads)
    idx = rs.choice(len(flat), 40, replace=False)
    eps = 1e-6
    for i in idx:
        fp, fm = flat.copy(), flat.copy()
        fp[i] += eps
        fm[i] -= eps
        lp = O.network_loss_and_grads(spec, O.unflatten(spec, fp), feats, seq_len, labels, label_len)[0]
        lm = O.network_loss_and_grads(spec, O.unflatten(spec, fm), feats, seq_len, labels, label_len)[0]
        assert gflat[i] == pytest.approx((lp - lm) / (2 * eps), abs=5e-8)


# ----------------------------------------------------------------- D3 index map on a toy
def test_stack_reshape_index_map_toy():
    """logits[t', b'] = O.flat_row(b'*2T + t') with O = stack(fw, bw) [2,B,T,H]:
    column b' < B/2 sees the fw outputs of utterance 2b' for t' < T."""
    B, T, H, C = 2, 3, 2, 2
    spec = O.ModelSpec(1, H, 1, True, 'stack_reshape', C)
    rs = np.random.RandomState(0)
    params = [rs.randn(*s) for _, s in spec.param_shapes()]
    feats = rs.randn(B, T, 1)
    seq = [T, T]
    logits, fc = O.network_forward(spec, params, feats, seq)
    of, _ = O.lstm_dir_forward(feats, seq, params[0], params[1], 1.0, False)
    ob, _ = O.lstm_dir_forward(feats, seq, params[2], params[3], 1.0, True)
    W, b = params[-2], params[-1]
    assert logits.shape == (2 * T, B, C)
    for bp in range(B):
        for tp in range(2 * T):
            r = bp * 2 * T + tp
            d, rem = divmod(r, B * T)
            bb, tt = divmod(rem, T)
            src = (of, ob)[d][bb, tt]
            np.testing.assert_allclose(logits[tp, bp], src @ W + b, atol=1e-13)
    # b'=0, t'<T  -> fw outputs of utterance 0;  b'=1, t'<T -> bw outputs of utterance 0
    np.testing.assert_allclose(logits[:T, 0], of[0] @ W + b, atol=1e-13)
    np.testing.assert_allclose(logits[:T, 1], ob[0] @ W + b, atol=1e-13)


# ----------------------------------------------------------------- Adam, DP, decode, helpers
def test_adam_tf_first_steps_by_hand():
    p, g = [np.array([1.0, -2.0])], [np.array([0.5, -0.25])]
    m, v = [np.zeros(2)], [np.zeros(2)]
    lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-8
    p1, m1, v1 = O.adam_tf(p, g, m, v, 1, lr)
    mm = (1 - b1) * g[0]
    vv = (1 - b2) * g[0] ** 2
    lr_t = lr * np.sqrt(1 - b2) / (1 - b1)
    np.testing.assert_allclose(p1[0], p[0] - lr_t * mm / (np.sqrt(vv) + eps), rtol=1e-15)
    # epsilon is NOT bias corrected: differs from torch.optim.Adam for tiny gradients
    tp = torch.tensor([1.0, -2.0], dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([tp], lr=lr, eps=eps)
    tp.grad = torch.tensor([1e-9, -1e-9], dtype=torch.float64)
    opt.step()
    q, _, _ = O.adam_tf(p, [np.array([1e-9, -1e-9])], m, v, 1, lr)
    assert abs(q[0][0] - float(tp[0])) > 1e-4


def test_data_parallel_equals_global_batch_for_concat():
    spec = O.ModelSpec(4, 3, 1, True, 'concat', 5)
    rs = np.random.RandomState(2)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 4, 8, seed=5, var_len=True, Lmin=1, Lmax=3)
    params = O.init_params(spec, seed=2)
    l1, _, g1, _ = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    l2, g2 = O.data_parallel_loss_and_grads(spec, params, feats, seq_len, labels, label_len, 2)
    assert l1 == pytest.approx(l2, rel=1e-13)
    for a, b in zip(g1, g2):
        np.testing.assert_allclose(a, b, atol=1e-13)


def test_data_parallel_stack_reshape_is_per_shard():
    """The D3 map depends on the shard's own B, so 2 towers != 1 tower for the literal net."""
    spec = O.ModelSpec(4, 3, 1, True, 'stack_reshape', 5)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 4, 8, seed=6, Lmin=1, Lmax=3)
    params = O.init_params(spec, seed=2)
    l1 = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)[0]
    l2, _ = O.data_parallel_loss_and_grads(spec, params, feats, seq_len, labels, label_len, 2)
    assert abs(l1 - l2) > 1e-6


def test_greedy_decode_and_ler():
    C = 4
    lg = np.full((6, 1, C), -5.0)
    for t, k in enumerate([0, 0, 3, 0, 1, 1]):
        lg[t, 0, k] = 5.0
    assert O.greedy_decode(lg, [6]) == [[0, 0, 1]]
    assert O.greedy_decode(lg, [2]) == [[0]]
    assert O.edit_distance([0, 0, 1], [0, 1]) == 1
    assert O.label_error_rate([[0, 0, 1]], np.array([[0, 1, 0]]), [2]) == pytest.approx(0.5)


def test_sparse_tuple_from_golden():
    """Golden from SURVEY §8c(4), captured by running the reference's utils.sparse_tuple_from."""
    i, v, s = O.sparse_tuple_from([[1, 2, 0], [3, 0, 0]], [2, 1])
    assert i.tolist() == [[0, 0], [0, 1], [1, 0]] and i.dtype == np.int64
    assert v.tolist() == [1, 2, 3] and v.dtype == np.int32
    assert s.tolist() == [2, 2] and s.dtype == np.int64


# ----------------------------------------------------------------- DeepSpeech dense stages
def _torch_deepspeech(spec, params, feats, seq_len, labels, label_len, masks):
    x = torch.tensor(feats, dtype=torch.float64).permute(1, 0, 2)
    B, T = feats.shape[0], feats.shape[1]
    pi = 0

    def dense(x, b, W, m, p):
        a = torch.clamp(torch.relu(x @ W + b), max=spec.relu_clip)
        return a * torch.tensor(m, dtype=torch.float64) / (1.0 - p)
    for i in range(len(spec.pre)):
        x = dense(x, params[pi], params[pi + 1], masks[i], spec.drop_p(i))
        pi += 2
    xb = x.permute(1, 0, 2)
    of = _torch_lstm_dir(xb, seq_len, params[pi], params[pi + 1], spec.forget_bias, False)
    ob = _torch_lstm_dir(xb, seq_len, params[pi + 2], params[pi + 3], spec.forget_bias, True)
    pi += 4
    x = torch.cat([of, ob], 2).permute(1, 0, 2)
    if spec.post:
        x = dense(x, params[pi], params[pi + 1], masks[len(spec.pre)], spec.drop_p(len(spec.pre)))
        pi += 2
    logits = x @ params[pi + 1] + params[pi]
    nll = torch.nn.functional.ctc_loss(torch.log_softmax(logits, -1), torch.tensor(labels), torch.tensor(seq_len),
                                       torch.tensor(label_len), blank=spec.num_classes - 1, reduction='none')
    return logits, nll.mean()


def test_deepspeech_family_matches_torch_autograd():
    spec = O.ModelSpec(6, 5, 1, True, 'concat', 5, pre=(7, 8, 10), post=9, relu_clip=1.5, dropout=(0.3, 0.2, 0.25, 0.4))
    names = [n for n, _ in spec.param_shapes()]
    assert names == ['b1', 'h1', 'b2', 'h2', 'b3', 'h3', 'l0/fw/kernel', 'l0/fw/bias', 'l0/bw/kernel', 'l0/bw/bias',
                     'b5', 'h5', 'b6', 'h6']                      # creation order of networks/deepspeech.py
    rs = np.random.RandomState(5)
    B, T = 3, 8
    seq_len = np.array([8, 6, 8])
    feats = rs.randn(B, T, 6)
    labels = rs.randint(0, 4, size=(B, 3))
    label_len = np.array([3, 2, 1])
    params = [p + 0.3 * rs.randn(*p.shape) for p in O.init_params(spec, seed=3)]
    drop = (1234, 7)
    masks = [O.dropout_mask(1234, 7, i, T, B, w, spec.drop_p(i)) for i, w in enumerate(list(spec.pre) + [spec.post])]
    assert 0.5 < masks[0].mean() < 0.9 and not masks[0].all()
    loss, nll, grads, logits = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len, drop=drop)
    tp = [torch.tensor(p, requires_grad=True) for p in params]
    tlogits, tloss = _torch_deepspeech(spec, tp, feats, seq_len, labels, label_len, masks)
    tloss.backward()
    np.testing.assert_allclose(logits, tlogits.detach().numpy(), atol=1e-12)
    assert loss == pytest.approx(float(tloss.detach()), rel=1e-12)
    for n, g, t in zip(names, grads, tp):
        np.testing.assert_allclose(g, t.grad.numpy(), atol=1e-11, err_msg=n)
    # without dropout the masks are all ones
    l0 = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)[0]
    assert l0 != loss


def test_dropout_mask_is_a_pure_function_of_its_key():
    a = O.dropout_mask(1, 2, 0, 4, 3, 5, 0.5)
    assert (a == O.dropout_mask(1, 2, 0, 4, 3, 5, 0.5)).all()
    assert (a != O.dropout_mask(1, 3, 0, 4, 3, 5, 0.5)).any() and (a != O.dropout_mask(1, 2, 1, 4, 3, 5, 0.5)).any()
    big = O.dropout_mask(9, 9, 2, 50, 8, 64, 0.05)
    assert abs(big.mean() - 0.95) < 0.01
