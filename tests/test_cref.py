"""The C/OpenMP restatement (oracle/cref/nasr_cref.c) against the fp64 NumPy oracle: it is the second CPU checker
and the timed cpu_baseline of bench.py, so it is pinned here on every net family at sizes the oracle does in seconds."""
import numpy as np
import pytest

from oracle import cref
from oracle import nasr_oracle as O

SPECS = [
    (O.ModelSpec(13, 24, 1, True, 'stack_reshape', 7), 4, 20),
    (O.ModelSpec(13, 24, 2, True, 'concat', 9), 6, 15),
    (O.ModelSpec(9, 20, 3, False, 'none', 5), 5, 12),
    (O.ModelSpec(9, 17, 1, True, 'concat', 29), 3, 25),
    (O.ModelSpec(5, 8, 1, False, 'none', 4), 1, 7),
]


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("spec,B,T", SPECS, ids=lambda v: str(v) if isinstance(v, int) else
                         f"L{v.num_layers}{'bi' if v.bidirectional else 'uni'}-{v.merge}")
def test_cref_matches_numpy_oracle(spec, B, T):
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=B * 7 + T, var_len=True, Lmin=1,
                                                      Lmax=max(1, T // 4))
    rs = np.random.RandomState(1)
    params = [(p + 0.05 * rs.randn(*p.shape)).astype(np.float32).astype(np.float64) for p in O.init_params(spec, seed=2)]
    lo, nllo, go, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    loss, nll, g, lg = cref.loss_and_grads(spec, O.flatten(params), feats, seq_len, labels, label_len, want_logits=True)
    np.testing.assert_allclose(lg, logits_o, atol=2e-5)
    assert loss == pytest.approx(lo, rel=1e-5)
    np.testing.assert_allclose(nll, nllo, rtol=1e-5, atol=1e-5)
    assert rel(g, O.flatten(go)) < 2e-5
    off = 0
    for (name, shp), t in zip(spec.param_shapes(), go):
        n = t.size
        assert rel(g[off:off + n], t.ravel()) < 1e-4, name
        off += n


DS_SPECS = [
    (O.ModelSpec(12, 16, 1, True, 'concat', 7, pre=(20, 17, 24), post=18, relu_clip=1.0, dropout=(0.2, 0.1, 0.3, 0.25)), 4, 13),
    (O.ModelSpec(10, 12, 1, True, 'concat', 6, pre=(16,), post=0, relu_clip=2.0, dropout=(0.5,)), 3, 11),
    (O.ModelSpec(10, 12, 2, False, 'none', 6, pre=(), post=14, relu_clip=0.7, dropout=(0.4,)), 4, 9),
]


@pytest.mark.parametrize("spec,B,T", DS_SPECS, ids=lambda v: str(v) if isinstance(v, int) else
                         f"pre{len(v.pre)}-post{v.post}")
@pytest.mark.parametrize("drop", [None, (4567, 3)], ids=['nodrop', 'drop'])
def test_cref_deepspeech_family_matches_numpy_oracle(spec, B, T, drop):
    """networks/deepspeech.py's family: clipped-ReLU dense stages with the hash-defined dropout masks, TF creation order of
    the variables (b1,h1,..., cells, b5,h5, b6,h6)."""
    rs = np.random.RandomState(1)
    params = [(p + 0.15 * rs.randn(*p.shape)).astype(np.float32).astype(np.float64) for p in O.init_params(spec, seed=2)]
    best = (-1.0, None)
    for seed in range(20):             # keep the comparison off the kinks of the clipped ReLU (see tests/test_gpu_deepspeech.py)
        batch = O.synth_batch(spec, B, T, seed=100 + seed, var_len=True, Lmin=1, Lmax=max(1, T // 4))
        m = O.deepspeech_kink_margin(spec, params, batch[0], batch[1], drop=drop)
        if m > best[0]:
            best = (m, batch)
    assert best[0] > 1e-5
    feats, seq_len, labels, label_len = best[1]
    lo, nllo, go, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len, drop=drop)
    loss, nll, g, lg = cref.loss_and_grads(spec, O.flatten(params), feats, seq_len, labels, label_len, want_logits=True,
                                           drop=drop)
    np.testing.assert_allclose(lg, logits_o, atol=2e-5)
    assert loss == pytest.approx(lo, rel=1e-5)
    off = 0
    for (name, shp), t in zip(spec.param_shapes(), go):
        n = t.size
        assert rel(g[off:off + n], t.ravel()) < 1e-4, name
        off += n
    assert off == g.size


def test_cref_infeasible_label():
    spec = O.ModelSpec(4, 8, 1, True, 'concat', 5)
    with pytest.raises(ValueError, match='Not enough time'):
        cref.loss_and_grads(spec, O.flatten(O.init_params(spec)), np.zeros((1, 2, 4), np.float32), [2], [[1, 1]], [2])


def test_cref_reports_threads():
    assert cref.num_threads() >= 1
