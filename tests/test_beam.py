"""CTC beam search (tf.nn.ctc_beam_search_decoder defaults, networks/tfnetwork.py:61-64): the library's host
C++ decoder against (a) exhaustive enumeration when the beam is wide enough to be exact, (b) the oracle's
independent Python restatement of TF's per-frame procedure on larger inputs, (c) TF's merge_repeated quirk.
Host code: runs without a GPU."""
import ctypes

import numpy as np
import pytest

from neuralasr_amd import _lib
from oracle import nasr_oracle as O


def lib_beam(logits_tm, seq_len, width=100, merge=True):
    lib = _lib.load()
    lg = np.ascontiguousarray(logits_tm, np.float32)
    Tp, B, C = lg.shape
    seq = np.ascontiguousarray(seq_len, np.int32)
    ids = np.zeros((B, Tp), np.int32)
    lens = np.zeros(B, np.int32)
    logp = np.zeros(B, np.float32)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    rc = lib.nasr_ctc_beam_search(lg.ctypes.data_as(fp), seq.ctypes.data_as(ip), B, Tp, C, width, int(merge),
                                  ids.ctypes.data_as(ip), lens.ctypes.data_as(ip), logp.ctypes.data_as(fp))
    assert rc == 0
    return [ids[b, :lens[b]].tolist() for b in range(B)], logp


@pytest.mark.parametrize("T,C,seed", [(1, 3, 0), (3, 3, 1), (4, 3, 2), (5, 3, 3), (4, 4, 4), (6, 3, 5)])
def test_wide_beam_is_exact(T, C, seed):
    rs = np.random.RandomState(seed)
    logits = rs.randn(T, 1, C) * 2.0
    best, logp, table = O.ctc_best_labelling_brute_force(logits[:, 0], C - 1)
    got, lp = lib_beam(logits, [T], width=1000, merge=False)
    assert got[0] == best
    assert lp[0] == pytest.approx(logp, abs=1e-5)
    ids_o, lp_o = O.ctc_beam_search(logits[:, 0], 1000, merge_repeated=False)
    assert ids_o == best and lp_o == pytest.approx(logp, abs=1e-10)


def test_matches_python_restatement_on_long_inputs():
    rs = np.random.RandomState(7)
    Tp, B, C = 60, 5, 9
    logits = (rs.randn(Tp, B, C) * 1.5).astype(np.float32)
    logits[:, :, C - 1] += 1.0                       # blank-heavy, like a trained CTC net
    seq = [60, 45, 1, 33, 60]
    for width, merge in ((100, True), (8, True), (8, False), (1, True)):
        got, lp = lib_beam(logits, seq, width, merge)
        for b in range(B):
            ids_o, lp_o = O.ctc_beam_search(logits[:seq[b], b].astype(np.float64), width, merge)
            assert got[b] == ids_o, (width, merge, b)
            assert lp[b] == pytest.approx(lp_o, abs=2e-3)


def test_merge_repeated_merges_across_blanks_like_tf():
    """TF's merge_repeated on the beam OUTPUT collapses 'a _ a' to 'a' (SURVEY.md Appendix A.6)."""
    C = 3                                            # labels 0,1; blank 2
    lg = np.full((3, 1, C), -9.0, np.float32)
    for t, k in enumerate([0, 2, 0]):
        lg[t, 0, k] = 9.0
    assert lib_beam(lg, [3], 100, merge=False)[0] == [[0, 0]]
    assert lib_beam(lg, [3], 100, merge=True)[0] == [[0]]
    assert O.greedy_decode(lg.astype(np.float64), [3]) == [[0, 0]]   # the greedy decoder keeps both


def test_width_one_equals_greedy_when_peaky():
    rs = np.random.RandomState(3)
    Tp, B, C = 40, 3, 6
    am = rs.randint(0, C, size=(Tp, B))
    lg = np.full((Tp, B, C), -8.0, np.float32)
    for t in range(Tp):
        for b in range(B):
            lg[t, b, am[t, b]] = 8.0
    got, _ = lib_beam(lg, [40, 40, 40], 100, merge=False)
    assert got == O.greedy_decode(lg.astype(np.float64), [40, 40, 40])


def test_bad_arguments():
    lib = _lib.load()
    assert lib.nasr_ctc_beam_search(None, None, 1, 1, 3, 100, 1, None, None, None) == _lib.NASR_ERR_ARG
