"""GPU: a randomised sweep of the CTC-facing shapes through the C ABI against the fp64 oracle (tf.nn.ctc_loss,
networks/tfnetwork.py:58-59): class counts from 3 to 31 (the default alpha / beta kernel of csrc/ctc.hip (2b)) and beyond (the
plain one), batches from 1 to 40, 1 to 130 frames (1 to 5 chunks of the emission ring, every phase of the 4-frame groups),
label lengths from 0 to the longest the frames allow, runs of identical labels (no skip transition), ragged lengths."""
import numpy as np
import pytest

from oracle import nasr_oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def feasible_len(lab, T):
    """labels the frames can hold: len + number of adjacent repeats <= T"""
    keep = []
    for v in lab:
        need = len(keep) + 1 + sum(1 for i in range(1, len(keep) + 1) if (keep + [v])[i] == (keep + [v])[i - 1])
        if need > T:
            break
        keep.append(v)
    return keep


@pytest.mark.parametrize('seed', range(28))
def test_random_shapes_match_the_oracle(seed):
    from neuralasr_amd.engine import Engine
    rs = np.random.RandomState(1000 + seed)
    C = int(rs.choice([3, 4, 6, 9, 17, 29, 31, 33, 40]))
    B = int(rs.choice([1, 2, 3, 5, 8, 16, 17, 40]))
    T = int(rs.choice([1, 2, 3, 4, 5, 6, 9, 31, 32, 33, 34, 36, 64, 65, 97, 130]))
    bi = bool(rs.randint(2))
    spec = O.ModelSpec(int(rs.randint(3, 12)), int(rs.choice([8, 16, 24])), 1, bi, 'concat' if bi else 'none', C)
    seq_len = rs.randint(1, T + 1, size=B).astype(np.int32)
    seq_len[rs.randint(B)] = T
    feats = rs.randn(B, T, spec.feature_size).astype(np.float32)
    labs = []
    for b in range(B):
        feats[b, seq_len[b]:] = 0
        style = rs.randint(4)
        want = [0, rs.randint(0, seq_len[b] + 1), seq_len[b], rs.randint(0, max(seq_len[b] // 3, 1) + 1)][style]
        if rs.randint(3) == 0:                       # runs of identical labels
            raw = np.repeat(rs.randint(0, C - 1, size=want // 2 + 1), 2)[:want]
        else:
            raw = rs.randint(0, C - 1, size=want)
        labs.append(feasible_len([int(v) for v in raw], int(seq_len[b])))
    Lmax = max(1, max(len(l) for l in labs))
    labels = np.zeros((B, Lmax), np.int32)
    label_len = np.array([len(l) for l in labs], np.int32)
    for b, l in enumerate(labs):
        labels[b, :len(l)] = l
    params = [p.astype(np.float32).astype(np.float64) * (1.0 + 4.0 * (seed % 3 == 2)) for p in O.init_params(spec, seed=seed)]
    e = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes)
    e.set_params(O.flatten(params))
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    lo, nllo, go, _ = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    assert loss == pytest.approx(lo, rel=3e-5, abs=1e-6)
    np.testing.assert_allclose(nll, nllo, rtol=3e-5, atol=2e-5)
    assert rel(grads, O.flatten(go)) < 1e-4
    again = e.loss_and_grads(feats, seq_len, labels, label_len)
    assert again[0] == loss
    np.testing.assert_array_equal(again[2], grads)
    e.close()
