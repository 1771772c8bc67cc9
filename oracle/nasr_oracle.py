"""CPU oracle (fp64 NumPy) for the NeuralASR CTC training hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``neuralasr_amd``) never does and fails loudly when the HIP library is absent.

PARITY STATUS: **parity unpinned against TensorFlow.**  The reference's
arithmetic lives in TensorFlow 1.x (unpinned version, ``requirements.txt:1-2``
does not even list it), which is not in this image, and the reference ships no
tests, golden vectors or fixtures for this path.  This file
restates the documented TF-1.x semantics the reference's call sites rely on and
is pinned instead by (tests/test_oracle_*.py):
  * brute-force CTC by alignment enumeration (``ctc_brute_force``),
  * central finite differences of this restatement,
  * an independent torch-CPU cross-check (autograd LSTM with TF gate order and
    ``torch.nn.functional.ctc_loss(blank=C-1)``),
  * host-side goldens produced by importing the reference's own host modules
    (tests/golden/make_host_goldens.py).

Reference call sites followed (paths relative to /root/reference):
  networks/bilstm_ctc_net.py:10-52   graph order, stack-reshape quirk (D3)
  networks/lstm_ctc_net.py:10-47     3x LSTMCell MultiRNNCell + dynamic_rnn
  networks/tfnetwork.py:58-59        ctc_loss + reduce_mean
  networks/tfnetwork.py:61-64        decoder (greedy variant named in the comment)
  networks/tfnetwork.py:72-140       tower split, gradient averaging, Adam
  utils.py:44-58                     sparse_tuple_from
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass

import numpy as np

NEG_INF = -np.inf


# --------------------------------------------------------------------------- model spec
@dataclass(frozen=True)
class ModelSpec:
    """Shape of one network.  The literal reference nets are
    ``ModelSpec(F, 500, 1, True, 'stack_reshape', C)`` (networks/bilstm_ctc_net.py:14-45) and
    ``ModelSpec(F, 500, 3, False, 'none', C)`` (networks/lstm_ctc_net.py:14-43)."""
    feature_size: int
    hidden: int
    num_layers: int
    bidirectional: bool
    merge: str            # 'stack_reshape' | 'concat' | 'none' (unidirectional)
    num_classes: int
    forget_bias: float = 1.0
    # DeepSpeech-style dense stages (networks/deepspeech.py:43-121): clipped-ReLU layers with dropout before the
    # LSTM stack (`pre` = their widths) and one between the stack and the logits (`post`, 0 = none).
    pre: tuple = ()
    post: int = 0
    relu_clip: float = 20.0
    dropout: tuple = ()          # drop probability per dense layer: len(pre) entries, then one for `post`

    @property
    def dirs(self):
        return 2 if self.bidirectional else 1

    @property
    def deepspeech(self):
        return bool(self.pre) or self.post > 0

    def layer_input(self, l):
        if l == 0:
            return self.pre[-1] if self.pre else self.feature_size
        return self.hidden * self.dirs

    def drop_p(self, i):
        """drop probability of dense layer i (0..len(pre)-1 = pre layers, len(pre) = post layer)."""
        return float(self.dropout[i]) if i < len(self.dropout) else 0.0

    @property
    def proj_in(self):
        if self.bidirectional and self.merge == 'concat':
            return 2 * self.hidden
        return self.hidden

    def logit_frames(self, T):
        """tf.reshape(logits, [batch_s, -1, C]) (networks/bilstm_ctc_net.py:45): the stacked
        (fw, bw) tuple doubles the row count, so the time axis becomes 2T."""
        if self.bidirectional and self.merge == 'stack_reshape':
            return 2 * T
        return T

    def param_shapes(self):
        """TF variable creation order: per layer (fw kernel, fw bias, bw kernel, bw bias) or
        (kernel, bias); then W, b.  kernel rows are [input ; h] (Appendix A.1)."""
        H = self.hidden
        out = []
        if self.deepspeech:
            # networks/deepspeech.py creates b_i before h_i: b1,h1,b2,h2,b3,h3, fw/bw cells, b5,h5,b6,h6
            w_in = self.feature_size
            for i, w in enumerate(self.pre):
                out.append((f'b{i + 1}', (w,)))
                out.append((f'h{i + 1}', (w_in, w)))
                w_in = w
        for l in range(self.num_layers):
            I = self.layer_input(l)
            if self.bidirectional:
                for d in ('fw', 'bw'):
                    out.append((f'l{l}/{d}/kernel', (I + H, 4 * H)))
                    out.append((f'l{l}/{d}/bias', (4 * H,)))
            else:
                out.append((f'l{l}/kernel', (I + H, 4 * H)))
                out.append((f'l{l}/bias', (4 * H,)))
        if self.deepspeech:
            w_in = self.proj_in
            if self.post:
                out.append(('b5', (self.post,)))
                out.append(('h5', (w_in, self.post)))
                w_in = self.post
            out.append(('b6', (self.num_classes,)))
            out.append(('h6', (w_in, self.num_classes)))
            return out
        out.append(('W', (self.proj_in, self.num_classes)))
        out.append(('b', (self.num_classes,)))
        return out

    def param_count(self):
        return int(sum(int(np.prod(s)) for _, s in self.param_shapes()))


def init_params(spec: ModelSpec, seed=1, dtype=np.float64):
    """Synthetic weights: glorot-uniform kernels, zero biases,
    W ~ N(0, 2/(Hin+C)) (xavier normal, networks/bilstm_ctc_net.py:35-36), b = 0.
    TF's own seeded init is not reproducible outside TF (Appendix A.1)."""
    rs = np.random.RandomState(seed)
    params = []
    for name, shp in spec.param_shapes():
        if name.endswith('kernel'):
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            params.append(rs.uniform(-lim, lim, size=shp).astype(dtype))
        elif name[0] in 'hb' and name[1:].isdigit():       # DeepSpeech dense stages: N(0, 0.046875) (deepspeech.py:25)
            if name in ('h1', 'h6'):                        # xavier-normal (deepspeech.py:47,120)
                params.append((rs.randn(*shp) * np.sqrt(2.0 / (shp[0] + shp[1]))).astype(dtype))
            else:
                params.append((rs.randn(*shp) * 0.046875).astype(dtype))
        elif name == 'W':
            params.append((rs.randn(*shp) * np.sqrt(2.0 / (shp[0] + shp[1]))).astype(dtype))
        else:
            params.append(np.zeros(shp, dtype))
    return params


def flatten(params):
    return np.concatenate([np.asarray(p).ravel() for p in params])


def unflatten(spec: ModelSpec, flat):
    out, off = [], 0
    for _, shp in spec.param_shapes():
        n = int(np.prod(shp))
        out.append(np.asarray(flat[off:off + n]).reshape(shp))
        off += n
    assert off == len(flat)
    return out


# --------------------------------------------------------------------------- LSTM
def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_dir_forward(x, seq_len, kernel, bias, forget_bias=1.0, reverse=False):
    """One direction of (bidirectional_)dynamic_rnn over BasicLSTMCell/LSTMCell.

    Appendix A.1: g = [x,h]·kernel + bias; i,j,f,o = split(g,4); c' = c·σ(f+fb) + σ(i)·tanh(j);
    h' = tanh(c')·σ(o).  A.2: past seq_len the output is zero and the state is carried.
    A.3: the bw direction runs over reverse_sequence(x, seq_len) and its output is reversed
    back, i.e. at step s row b consumes/produces frame seq_len[b]-1-s.
    x: [B,T,I] -> out [B,T,H], cache."""
    x = np.asarray(x, np.float64)
    B, T, I = x.shape
    H = kernel.shape[1] // 4
    seq_len = np.asarray(seq_len, np.int64)
    out = np.zeros((B, T, H))
    h = np.zeros((B, H))
    c = np.zeros((B, H))
    rows = np.arange(B)
    steps = []
    for s in range(T):
        valid = s < seq_len
        tb = (seq_len - 1 - s) if reverse else np.full(B, s)
        tb = np.where(valid, tb, 0)
        xt = x[rows, tb] * valid[:, None]
        z = np.concatenate([xt, h], 1)
        g = z @ kernel + bias
        si = _sig(g[:, 0:H])
        tj = np.tanh(g[:, H:2 * H])
        sf = _sig(g[:, 2 * H:3 * H] + forget_bias)
        so = _sig(g[:, 3 * H:4 * H])
        c_new = c * sf + si * tj
        tc = np.tanh(c_new)
        h_new = tc * so
        steps.append((valid, tb, z, si, tj, sf, so, c.copy(), tc))
        vm = valid[:, None]
        out[rows[valid], tb[valid]] = h_new[valid]
        c = np.where(vm, c_new, c)
        h = np.where(vm, h_new, h)
    cache = dict(steps=steps, kernel=kernel, I=I, H=H, B=B, T=T)
    return out, cache


def lstm_dir_backward(cache, dout):
    """BPTT for lstm_dir_forward.  dout: [B,T,H] -> (dx [B,T,I], dkernel, dbias).
    Masked steps pass dh/dc through and add nothing to the parameters (A.2)."""
    K, I, H, B, T = cache['kernel'], cache['I'], cache['H'], cache['B'], cache['T']
    dx = np.zeros((B, T, I))
    dK = np.zeros_like(K)
    db = np.zeros(4 * H)
    dh = np.zeros((B, H))
    dc = np.zeros((B, H))
    rows = np.arange(B)
    for s in reversed(range(T)):
        valid, tb, z, si, tj, sf, so, c_prev, tc = cache['steps'][s]
        vm = valid[:, None]
        dh_tot = dh + dout[rows, tb] * vm
        do = dh_tot * tc * so * (1 - so)
        dc_tot = dc + dh_tot * so * (1 - tc * tc)
        di = dc_tot * tj * si * (1 - si)
        dj = dc_tot * si * (1 - tj * tj)
        df = dc_tot * c_prev * sf * (1 - sf)
        dg = np.concatenate([di, dj, df, do], 1) * vm
        dK += z.T @ dg
        db += dg.sum(0)
        dz = dg @ K.T
        dx[rows[valid], tb[valid]] += dz[valid, :I]
        dh = np.where(vm, dz[:, I:], dh)
        dc = np.where(vm, dc_tot * sf, dc)
    return dx, dK, db


# --------------------------------------------------------------------------- CTC
def _lse(*xs):
    m = np.max(np.stack(xs), axis=0)
    safe = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide='ignore'):
        return np.where(np.isfinite(m), safe + np.log(sum(np.exp(x - safe) for x in xs)), NEG_INF)


def _shift(a, k):
    """out[s] = a[s-k] (k>0) or a[s+|k|] (k<0), -inf where out of range."""
    out = np.full_like(a, NEG_INF)
    n = len(a)
    if k > 0 and k < n:
        out[k:] = a[:n - k]
    elif k < 0 and -k < n:
        out[:n + k] = a[-k:]
    return out


def log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def ctc_feasible(label, T):
    """TF raises "Not enough time for target transition sequence" when
    L + #adjacent-repeats > seq_len (Appendix A.4)."""
    label = list(label)
    rep = sum(1 for a, b in zip(label[:-1], label[1:]) if a == b)
    return len(label) + rep <= T


def ctc_single(logits, label, blank):
    """tf.nn.ctc_loss for one utterance (Appendix A.4): logits [T,C] unnormalised, label ids.
    Returns (nll, dnll/dlogits [T,C], alpha, beta).  beta excludes the emission at t."""
    logits = np.asarray(logits, np.float64)
    T, C = logits.shape
    label = np.asarray(label, np.int64)
    L = len(label)
    if not ctc_feasible(label, T):
        raise ValueError('Not enough time for target transition sequence '
                         f'(required: {L + sum(label[:-1] == label[1:])}, available: {T})')
    S = 2 * L + 1
    ext = np.full(S, blank, np.int64)
    ext[1::2] = label
    lp = log_softmax(logits)
    lpe = lp[:, ext]                      # [T,S]
    skip = np.zeros(S, bool)              # may come from s-2
    skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
    alpha = np.full((T, S), NEG_INF)
    alpha[0, 0] = lpe[0, 0]
    if S > 1:
        alpha[0, 1] = lpe[0, 1]
    for t in range(1, T):
        a = alpha[t - 1]
        a1 = _shift(a, 1)
        a2 = np.where(skip, _shift(a, 2), NEG_INF)
        alpha[t] = lpe[t] + _lse(a, a1, a2)
    beta = np.full((T, S), NEG_INF)
    beta[T - 1, S - 1] = 0.0
    if S > 1:
        beta[T - 1, S - 2] = 0.0
    skip_f = np.zeros(S, bool)            # may go to s+2
    skip_f[:-2] = skip[2:]
    for t in range(T - 2, -1, -1):
        bb = beta[t + 1] + lpe[t + 1]
        b1 = _shift(bb, -1)
        b2 = np.where(skip_f, _shift(bb, -2), NEG_INF)
        beta[t] = _lse(bb, b1, b2)
    ab = alpha + beta
    m = ab[0].max()
    logp = m + np.log(np.exp(ab[0] - m).sum())
    post = np.zeros((T, C))
    w = np.exp(ab - logp)
    for s in range(S):
        post[:, ext[s]] += w[:, s]
    grad = np.exp(lp) - post
    return -logp, grad, alpha, beta


def ctc_loss_and_grad(logits_tm, labels, label_len, seq_len, blank=None):
    """Batch CTC (networks/tfnetwork.py:58-59 before the reduce_mean).
    logits_tm: [T',B,C] time-major; labels [B,Lmax] padded; only t < seq_len[b] is read.
    Returns (nll [B], dnll/dlogits [T',B,C] with zero rows for t >= seq_len[b])."""
    logits_tm = np.asarray(logits_tm, np.float64)
    Tp, B, C = logits_tm.shape
    blank = C - 1 if blank is None else blank
    nll = np.zeros(B)
    grad = np.zeros_like(logits_tm)
    for b in range(B):
        Tb = int(seq_len[b])
        lab = np.asarray(labels[b][:int(label_len[b])], np.int64)
        nll[b], g, _, _ = ctc_single(logits_tm[:Tb, b], lab, blank)
        grad[:Tb, b] = g
    return nll, grad


def ctc_brute_force(logits, label, blank):
    """Known-answer CTC: -log of the summed probability of every length-T alignment that
    collapses (merge repeats, drop blanks) to ``label``.  Exponential; T<=7, C<=4 only."""
    logits = np.asarray(logits, np.float64)
    T, C = logits.shape
    p = np.exp(log_softmax(logits))
    tot = 0.0
    label = list(int(v) for v in label)
    for path in itertools.product(range(C), repeat=T):
        col, prev = [], None
        for k in path:
            if k != prev and k != blank:
                col.append(k)
            prev = k
        if col == label:
            pr = 1.0
            for t, k in enumerate(path):
                pr *= p[t, k]
            tot += pr
    return -np.log(tot) if tot > 0 else np.inf


def greedy_decode(logits_tm, seq_len, blank=None):
    """tf.nn.ctc_greedy_decoder(merge_repeated=True) (Appendix A.6; the decoder the comment at
    networks/tfnetwork.py:62-63 names): per-frame argmax, collapse repeats, drop blanks."""
    Tp, B, C = logits_tm.shape
    blank = C - 1 if blank is None else blank
    out = []
    for b in range(B):
        am = np.argmax(logits_tm[:int(seq_len[b]), b], axis=-1)
        ids, prev = [], -1
        for k in am:
            if k != prev and k != blank:
                ids.append(int(k))
            prev = k
        out.append(ids)
    return out


def ctc_beam_search(logits, beam_width=100, merge_repeated=True, blank=None):
    """tf.nn.ctc_beam_search_decoder for one utterance (Appendix A.6; networks/tfnetwork.py:61-64 uses the
    defaults beam_width=100, top_paths=1, merge_repeated=True).  logits [T,C] -> (ids, log_prob of the best
    beam).  Restated from TF 1.x's documented per-frame procedure: every live prefix keeps
    (p_blank, p_label, p_total); per frame first the probabilities of the live prefixes are advanced, then each
    is grown by one label while the grown prefix can still enter the `beam_width` best.  Unpinned against TF;
    pinned by exhaustive enumeration when the beam is wide enough (tests/test_beam.py)."""
    logits = np.asarray(logits, np.float64)
    T, C = logits.shape
    blank = C - 1 if blank is None else blank
    NEG = -np.inf

    def lse2(a, b):
        if a == NEG:
            return b
        if b == NEG:
            return a
        m = max(a, b)
        return m + np.log(np.exp(a - m) + np.exp(b - m))

    class Node:
        __slots__ = ('parent', 'label', 'old', 'new', 'kids')

        def __init__(self, parent, label):
            self.parent, self.label, self.kids = parent, label, {}
            self.old = [NEG, NEG, NEG]      # total, blank, label
            self.new = [NEG, NEG, NEG]

    root = Node(None, -1)
    root.new = [0.0, 0.0, NEG]
    leaves = [root]

    def push(n):
        if len(leaves) < beam_width:
            leaves.append(n)
            return
        i = min(range(len(leaves)), key=lambda k: leaves[k].new[0])
        if n.new[0] > leaves[i].new[0]:
            leaves[i] = n

    def candidate(p):
        return p[0] > NEG and (len(leaves) < beam_width or p[0] > min(l.new[0] for l in leaves))

    for t in range(T):
        lp = log_softmax(logits[t])
        branches = sorted(leaves, key=lambda n: -n.new[0])
        leaves = []
        for n in branches:
            n.old = list(n.new)
        for n in branches:
            if n.parent is not None:
                if n.parent.new[0] != NEG:
                    prev = n.parent.old[1] if n.label == n.parent.label else n.parent.old[0]
                    n.new[2] = lse2(n.new[2], prev)
                n.new[2] += lp[n.label]
            n.new[1] = n.old[0] + lp[blank]
            n.new[0] = lse2(n.new[1], n.new[2])
            push(n)
        for n in branches:
            if not candidate(n.old):
                continue
            for lab in range(C):
                if lab == blank:
                    continue
                c = n.kids.get(lab)
                if c is None:
                    c = n.kids[lab] = Node(n, lab)
                if c.new[0] != NEG:
                    continue
                prev = n.old[1] if lab == n.label else n.old[0]
                c.new = [lp[lab] + prev, NEG, lp[lab] + prev]
                if candidate(c.new):
                    if len(leaves) == beam_width:
                        i = min(range(len(leaves)), key=lambda k: leaves[k].new[0])
                        leaves[i].new = [NEG, NEG, NEG]
                    push(c)
                else:
                    c.old = [NEG, NEG, NEG]
                    c.new = [NEG, NEG, NEG]
    best = max(leaves, key=lambda n: n.new[0])
    seq, prev, n = [], -1, best
    while n.parent is not None:
        if not merge_repeated or n.label != prev:
            seq.append(n.label)
        prev = n.label
        n = n.parent
    return seq[::-1], best.new[0]


def ctc_best_labelling_brute_force(logits, blank):
    """argmax over label sequences of the summed alignment probability (exponential; tiny cases only)."""
    logits = np.asarray(logits, np.float64)
    T, C = logits.shape
    p = np.exp(log_softmax(logits))
    tot = {}
    for path in itertools.product(range(C), repeat=T):
        col, prev = [], None
        for k in path:
            if k != prev and k != blank:
                col.append(k)
            prev = k
        pr = 1.0
        for t, k in enumerate(path):
            pr *= p[t, k]
        tot[tuple(col)] = tot.get(tuple(col), 0.0) + pr
    best = max(tot, key=tot.get)
    return list(best), float(np.log(tot[best])), tot


def edit_distance(hyp, truth):
    """Levenshtein distance (tf.edit_distance core, Appendix A.7)."""
    n, m = len(hyp), len(truth)
    d = list(range(m + 1))
    for i in range(1, n + 1):
        prev, d[0] = d[0], i
        for j in range(1, m + 1):
            cur = d[j]
            d[j] = min(d[j] + 1, d[j - 1] + 1, prev + (hyp[i - 1] != truth[j - 1]))
            prev = cur
    return d[m]


def label_error_rate(hyps, labels, label_len):
    """mean over the batch of edit_distance/len(truth) (networks/tfnetwork.py:66-70)."""
    v = []
    for h, lab, n in zip(hyps, labels, label_len):
        t = list(int(x) for x in lab[:int(n)])
        if t:
            v.append(edit_distance(list(h), t) / len(t))
        else:   # TF: empty truth -> inf for a non-empty hypothesis, 0 for an empty one
            v.append(np.inf if len(h) else 0.0)
    return float(np.mean(v))


# --------------------------------------------------------------------------- dense stages (DeepSpeech)
def dropout_mask(seed, counter, layer, T, B, W, p):
    """Keep-mask [T,B,W] of tf.nn.dropout(x, 1-p) (networks/deepspeech.py:50,59,68,113).  TF's random stream cannot be
    reproduced, so the mask is DEFINED by a counter-based hash both sides compute (oracle here, HIP in
    neuralasr_amd/csrc/dense.hip): element (t,b,j) of dense layer `layer` on forward pass number `counter` is kept iff
    lowbias32(idx ^ key) >> 8 >= floor(p * 2^24), idx = (t*B + b)*W + j."""
    if p <= 0.0:
        return np.ones((T, B, W), bool)
    idx = np.arange(T * B * W, dtype=np.uint64).astype(np.uint32)
    key = np.uint32((int(seed) + 0x9E3779B9 * (int(layer) + 1) + 0x85EBCA6B * int(counter)) & 0xFFFFFFFF)
    x = idx ^ key
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * 0x7FEB352D & 0xFFFFFFFF).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x.astype(np.uint64) * 0x846CA68B & 0xFFFFFFFF).astype(np.uint32)
    x ^= x >> np.uint32(16)
    thr = np.uint32(int(np.floor(p * (1 << 24))))
    return ((x >> np.uint32(8)) >= thr).reshape(T, B, W)


def dense_forward(x, W, b, clip, mask, p):
    """min(relu(x W + b), clip) then tf.nn.dropout: kept elements scaled by 1/(1-p).  x [..., I] -> [..., O]."""
    z = x @ W + b
    a = np.minimum(np.maximum(z, 0.0), clip)
    return a * mask / (1.0 - p), z


def dense_backward(dy, x, z, W, clip, mask, p):
    da = dy * mask / (1.0 - p)
    dz = da * ((z > 0.0) & (z < clip))
    x2, dz2 = x.reshape(-1, x.shape[-1]), dz.reshape(-1, dz.shape[-1])
    return (dz2 @ W.T).reshape(x.shape), x2.T @ dz2, dz2.sum(0)


# --------------------------------------------------------------------------- network
def network_forward(spec: ModelSpec, params, feats, seq_len, drop=None):
    if spec.deepspeech:
        return _deepspeech_forward(spec, params, feats, seq_len, drop)
    """create_network up to the time-major logits (networks/bilstm_ctc_net.py:17-48,
    networks/lstm_ctc_net.py:17-43).  feats [B,T,F] batch-major -> logits [T',B,C]."""
    feats = np.asarray(feats, np.float64)
    B, T, _ = feats.shape
    H, C = spec.hidden, spec.num_classes
    p = list(params)
    W, bproj = np.asarray(p[-2], np.float64), np.asarray(p[-1], np.float64)
    x = feats
    caches = []
    pi = 0
    for l in range(spec.num_layers):
        if spec.bidirectional:
            of, cf = lstm_dir_forward(x, seq_len, np.asarray(p[pi], np.float64), np.asarray(p[pi + 1], np.float64),
                                      spec.forget_bias, False)
            ob, cb = lstm_dir_forward(x, seq_len, np.asarray(p[pi + 2], np.float64), np.asarray(p[pi + 3], np.float64),
                                      spec.forget_bias, True)
            pi += 4
            caches.append((cf, cb))
            last = (of, ob)
            x = np.concatenate([of, ob], 2)
        else:
            o, c = lstm_dir_forward(x, seq_len, np.asarray(p[pi], np.float64), np.asarray(p[pi + 1], np.float64),
                                    spec.forget_bias, False)
            pi += 2
            caches.append((c,))
            last = (o,)
            x = o
    if spec.bidirectional and spec.merge == 'stack_reshape':
        # tf.reshape(outputs, [-1, H]) on the (fw, bw) tuple packs to [2,B,T,H]
        flat = np.stack(last, 0).reshape(-1, H)
    else:
        flat = x.reshape(-1, spec.proj_in)
    logits = (flat @ W + bproj).reshape(B, -1, C).transpose(1, 0, 2)
    return logits, dict(caches=caches, flat=flat, B=B, T=T)


def _deepspeech_forward(spec, params, feats, seq_len, drop):
    """networks/deepspeech.py:35-127: time-major dense stages -> BiLSTM (concat) -> dense -> logits [T,B,C].
    `drop` = (seed, counter) of the dropout masks, or None for no dropout."""
    feats = np.asarray(feats, np.float64)
    B, T, _ = feats.shape
    p = [np.asarray(q, np.float64) for q in params]
    x = feats.transpose(1, 0, 2)                      # [T,B,F]
    pi, dense = 0, []

    def mask_for(i, W):
        pr = spec.drop_p(i) if drop is not None else 0.0
        m = dropout_mask(drop[0], drop[1], i, T, B, W, pr) if pr > 0 else np.ones((T, B, W), bool)
        return m, pr
    for i, w in enumerate(spec.pre):
        b_, W_ = p[pi], p[pi + 1]
        pi += 2
        m, pr = mask_for(i, w)
        y, z = dense_forward(x, W_, b_, spec.relu_clip, m, pr)
        dense.append((x, z, W_, m, pr))
        x = y
    xb = x.transpose(1, 0, 2)                         # batch-major for the LSTM helpers
    caches = []
    for l in range(spec.num_layers):
        if spec.bidirectional:
            of, cf = lstm_dir_forward(xb, seq_len, p[pi], p[pi + 1], spec.forget_bias, False)
            ob, cb = lstm_dir_forward(xb, seq_len, p[pi + 2], p[pi + 3], spec.forget_bias, True)
            pi += 4
            caches.append((cf, cb))
            xb = np.concatenate([of, ob], 2)
        else:
            o, c = lstm_dir_forward(xb, seq_len, p[pi], p[pi + 1], spec.forget_bias, False)
            pi += 2
            caches.append((c,))
            xb = o
    x = xb.transpose(1, 0, 2)                         # [T,B,proj_in]
    post = None
    if spec.post:
        b_, W_ = p[pi], p[pi + 1]
        pi += 2
        m, pr = mask_for(len(spec.pre), spec.post)
        y, z = dense_forward(x, W_, b_, spec.relu_clip, m, pr)
        post = (x, z, W_, m, pr)
        x = y
    b6, W6 = p[pi], p[pi + 1]
    logits = x @ W6 + b6                              # [T,B,C], already time-major
    return logits, dict(caches=caches, dense=dense, post=post, last=x, W6=W6, B=B, T=T, pi_lstm=2 * len(spec.pre))


def _deepspeech_loss_and_grads(spec, params, feats, seq_len, labels, label_len, drop):
    logits, fc = _deepspeech_forward(spec, params, feats, seq_len, drop)
    B, T = fc['B'], fc['T']
    H = spec.hidden
    nll, dlog = ctc_loss_and_grad(logits, labels, label_len, seq_len)
    loss = float(nll.mean())
    dlog = dlog / B
    grads = [None] * len(params)
    last = fc['last']
    grads[-1] = last.reshape(-1, last.shape[-1]).T @ dlog.reshape(-1, spec.num_classes)   # h6
    grads[-2] = dlog.reshape(-1, spec.num_classes).sum(0)                                # b6
    dx = dlog @ fc['W6'].T
    pi = len(params) - 2
    if spec.post:
        x, z, W_, m, pr = fc['post']
        dx, dW, db = dense_backward(dx, x, z, W_, spec.relu_clip, m, pr)
        pi -= 2
        grads[pi], grads[pi + 1] = db, dW
    dxb = dx.transpose(1, 0, 2)                        # batch-major [B,T,proj_in]
    for l in reversed(range(spec.num_layers)):
        if spec.bidirectional:
            pi -= 4
            dxf, grads[pi], grads[pi + 1] = lstm_dir_backward(fc['caches'][l][0], dxb[:, :, :H])
            dxr, grads[pi + 2], grads[pi + 3] = lstm_dir_backward(fc['caches'][l][1], dxb[:, :, H:])
            dxb = dxf + dxr
        else:
            pi -= 2
            dxb, grads[pi], grads[pi + 1] = lstm_dir_backward(fc['caches'][l][0], dxb)
    dx = dxb.transpose(1, 0, 2)
    for i in reversed(range(len(spec.pre))):
        x, z, W_, m, pr = fc['dense'][i]
        dx, dW, db = dense_backward(dx, x, z, W_, spec.relu_clip, m, pr)
        grads[2 * i], grads[2 * i + 1] = db, dW
    return loss, nll, grads, logits


def deepspeech_kink_margin(spec, params, feats, seq_len, drop=None):
    """Smallest distance of any dense-stage pre-activation of a live frame to a kink of the clipped ReLU (0 or relu_clip),
    relative to max(1, |z|).  An fp32 implementation reproduces the oracle's ReLU masks - and with them its gradients to
    better than one mask element - only when this margin exceeds its own forward rounding (~1e-6): tests pick inputs
    with a clear margin instead of comparing across a discontinuity."""
    _, fc = _deepspeech_forward(spec, params, feats, seq_len, drop)
    T, B = fc['T'], fc['B']
    live = (np.arange(T)[:, None] < np.asarray(seq_len)[None, :])[:, :, None]          # [T,B,1]
    worst = np.inf
    for (x, z, W_, m, pr) in list(fc['dense']) + ([fc['post']] if fc['post'] is not None else []):
        d = np.minimum(np.abs(z), np.abs(z - spec.relu_clip)) / np.maximum(1.0, np.abs(z))
        d = np.where(live & m, d, np.inf)
        worst = min(worst, float(d.min()))
    return worst


def network_loss_and_grads(spec: ModelSpec, params, feats, seq_len, labels, label_len, drop=None):
    """loss = reduce_mean(ctc_loss) (networks/tfnetwork.py:59) and d loss / d every variable,
    in TF variable order.  Returns (loss, nll[B], grads list, logits)."""
    if spec.deepspeech:
        return _deepspeech_loss_and_grads(spec, params, feats, seq_len, labels, label_len, drop)
    logits, fc = network_forward(spec, params, feats, seq_len)
    B, T = fc['B'], fc['T']
    H, C = spec.hidden, spec.num_classes
    nll, dlog = ctc_loss_and_grad(logits, labels, label_len, seq_len)
    loss = float(nll.mean())
    dlog = dlog / B
    dflat_logits = dlog.transpose(1, 0, 2).reshape(-1, C)
    W = np.asarray(params[-2], np.float64)
    dW = fc['flat'].T @ dflat_logits
    db = dflat_logits.sum(0)
    dflat = dflat_logits @ W.T
    grads = [None] * len(params)
    grads[-2], grads[-1] = dW, db
    if spec.bidirectional and spec.merge == 'stack_reshape':
        d = dflat.reshape(2, B, T, H)
        dlast = (d[0], d[1])
    elif spec.bidirectional:
        d = dflat.reshape(B, T, 2 * H)
        dlast = (d[:, :, :H], d[:, :, H:])
    else:
        dlast = (dflat.reshape(B, T, H),)
    pi = len(params) - 2
    for l in reversed(range(spec.num_layers)):
        if spec.bidirectional:
            pi -= 4
            dxf, grads[pi], grads[pi + 1] = lstm_dir_backward(fc['caches'][l][0], dlast[0])
            dxb, grads[pi + 2], grads[pi + 3] = lstm_dir_backward(fc['caches'][l][1], dlast[1])
            dx = dxf + dxb
            dlast = (dx[:, :, :H], dx[:, :, H:]) if l > 0 else None
        else:
            pi -= 2
            dx, grads[pi], grads[pi + 1] = lstm_dir_backward(fc['caches'][l][0], dlast[0])
            dlast = (dx,)
    return loss, nll, grads, logits


# --------------------------------------------------------------------------- optimiser / DP
def adam_tf(params, grads, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (Appendix A.5): lr_t = lr·sqrt(1-β2^t)/(1-β1^t); ε is added to
    sqrt(v) un-corrected.  ``step`` is t (1 for the first update).  Returns new (params,m,v)."""
    lr_t = lr * np.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    np_, nm, nv = [], [], []
    for p, g, mm, vv in zip(params, grads, m, v):
        mm = beta1 * mm + (1 - beta1) * g
        vv = beta2 * vv + (1 - beta2) * g * g
        np_.append(p - lr_t * mm / (np.sqrt(vv) + eps))
        nm.append(mm)
        nv.append(vv)
    return np_, nm, nv


def shard_slices(global_batch, n):
    """tf.split(v, num_gpus) on axis 0 (networks/tfnetwork.py:101): equal contiguous shards."""
    if global_batch % n:
        raise ValueError(f'global batch {global_batch} does not split evenly over {n} towers')
    per = global_batch // n
    return [slice(i * per, (i + 1) * per) for i in range(n)]


def data_parallel_loss_and_grads(spec, params, feats, seq_len, labels, label_len, n):
    """make_parallel + average_gradients (networks/tfnetwork.py:72-140): each tower gets a
    contiguous shard padded to the GLOBAL max T, computes its shard-mean loss and gradients
    (the D3 index map applies per shard); loss and every gradient are the mean over towers."""
    seq_len = np.asarray(seq_len)
    labels = np.asarray(labels)
    label_len = np.asarray(label_len)
    losses, gsum = [], None
    for sl in shard_slices(feats.shape[0], n):
        loss, _, g, _ = network_loss_and_grads(spec, params, feats[sl], seq_len[sl], labels[sl], label_len[sl])
        losses.append(loss)
        gsum = g if gsum is None else [a + b for a, b in zip(gsum, g)]
    return float(np.mean(losses)), [g / n for g in gsum]


def sparse_tuple_from(sequences, output_lengths):
    """utils.py:44-58 restated: dense padded labels + lengths -> COO (indices, values, shape)."""
    indices, values = [], []
    for n, seq in enumerate(sequences):
        l = int(output_lengths[n])
        indices.extend((n, j) for j in range(l))
        values.extend(int(x) for x in seq[:l])
    indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
    values = np.asarray(values, dtype=np.int32)
    shape = np.asarray([len(sequences), indices[:, 1].max() + 1], dtype=np.int64)
    return indices, values, shape


# --------------------------------------------------------------------------- synthetic workload
def synth_batch(spec: ModelSpec, B, T, seed=1234, var_len=False, Lmin=None, Lmax=None):
    """BASELINE.md §4 synthetic batch: features N(0,1) zeroed past seq_len, labels U{1..C-2},
    L ~ U{Lmin..Lmax} (40..80 at T=500, scaled down for small T), seq_len = T or U{T/2..T} sorted."""
    rs = np.random.RandomState(seed)
    C = spec.num_classes
    if var_len:
        seq_len = np.sort(rs.randint(max(T // 2, 1), T + 1, size=B)).astype(np.int32)
        seq_len[-1] = T
    else:
        seq_len = np.full(B, T, np.int32)
    feats = rs.randn(B, T, spec.feature_size).astype(np.float32)
    for b in range(B):
        feats[b, seq_len[b]:] = 0
    if Lmax is None:
        Lmax = max(1, min(80, T * 80 // 500))
    if Lmin is None:
        Lmin = max(1, Lmax // 2)
    label_len = rs.randint(Lmin, Lmax + 1, size=B).astype(np.int32)
    label_len = np.minimum(label_len, np.maximum(seq_len // 2, 1)).astype(np.int32)
    labels = np.zeros((B, int(label_len.max())), np.int32)
    for b in range(B):
        labels[b, :label_len[b]] = rs.randint(1, max(C - 1, 2), size=label_len[b])
    return feats, seq_len, labels, label_len
