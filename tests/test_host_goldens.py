"""Host-side modules (config / symbols / dataset / utils) against goldens captured by RUNNING the
reference's own host modules (tests/golden/make_host_goldens.py) and the values SURVEY.md §8c(4) records."""
import json
import os

import numpy as np
import pytest

from neuralasr_amd import utils
from neuralasr_amd.config import Config
from neuralasr_amd.dataset import DataSet
from neuralasr_amd.symbols import Symbols

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, 'golden', 'host_goldens.json')))
BATCHES = np.load(os.path.join(HERE, 'golden', 'host_batches.npz'))
SAMPLES = os.path.join(HERE, 'golden', 'sample_set')


@pytest.fixture
def toy_config(tmp_path):
    """The config the reference parsed, re-rooted at this checkout (absolute paths differ per box)."""
    text = open(os.path.join(SAMPLES, 'toy.config')).read()
    lines = []
    for ln in text.splitlines():
        if ln.startswith('output='):
            ln = 'output=' + SAMPLES
        if ln.startswith('model_dir='):
            ln = 'model_dir=' + str(tmp_path / 'model')
        lines.append(ln)
    p = tmp_path / 'toy.config'
    p.write_text('\n'.join(lines) + '\n')
    return str(p)


def test_symbols_roundtrip_matches_reference(tmp_path):
    s = Symbols(0)
    s.insert_padding()
    for c in 'ab_':
        s.insert_sym(c)
    s.insert_blank()
    f = tmp_path / 'symbols'
    s.write(str(f))
    assert f.read_text() == GOLD['symbols_file_text'] == '<blank> 4\n<padding> 0\n_ 3\na 1\nb 2\n'
    t = Symbols(0, str(f))
    assert t.counter == GOLD['symbols_counter'] == 5
    assert t.get_padding_id() == GOLD['symbols_padding_id'] == 0
    assert t.sym_to_id == GOLD['symbols_sym_to_id']
    assert t.convert_to_str(GOLD['symbols_convert']['ids']) == GOLD['symbols_convert']['str']
    assert s.insert_sym('a') == 1 and s.counter == 5       # re-insert is a lookup


def test_symbols_label_context_strips_context():
    s = Symbols(1)
    for tri in ('^ab', 'abc', 'bc^'):
        s.insert_sym(tri)
    assert s.convert_to_str([0, 1, 2]) == 'abc'


def test_config_derived_fields_match_reference(toy_config):
    c = Config(toy_config, True)
    g = GOLD['config']
    for k in ['samplerate', 'numcep', 'numcontext', 'rand_shift', 'feature_size', 'batch_size', 'epochs',
              'learningrate', 'start_step', 'report_step', 'num_gpus', 'label_context', 'punc_regex', 'network',
              'start_marker', 'end_marker']:
        assert getattr(c, k) == g[k], k
    assert c.batch_size == 4 and c.feature_size == 9          # 2 per GPU x 2 GPUs; (2*1+1)*3
    assert os.path.basename(c.sym_file) == g['sym_file_basename']
    assert os.path.basename(c.train_input) == g['train_input_basename']
    assert os.path.basename(c.test_input) == g['test_input_basename']
    assert c.symbols.counter == g['symbols_counter']


def test_config_missing_key_raises(tmp_path):
    p = tmp_path / 'bad.config'
    p.write_text('[Parameters]\nsamplerate=8000\n[Train]\n[Test]\n[MFCC Featurizer]\n')
    with pytest.raises(KeyError):
        Config(str(p))


def test_config_loads_hip_network_for_reference_dotted_name(toy_config):
    c = Config(toy_config, True)
    import importlib
    mod = importlib.import_module('neuralasr_amd.networks.bilstm_ctc_net')
    assert c.network == 'networks.bilstm_ctc_net.BiLstmCTCNet'
    assert hasattr(mod, 'BiLstmCTCNet')
    c.network = 'networks.nothing.Missing'
    with pytest.raises(ImportError):
        c.load_network()


def test_dataset_batches_match_reference(toy_config):
    c = Config(toy_config, True)
    ds = DataSet(c.train_input, c)
    g = GOLD['dataset']
    assert ds.get_num_of_sample() == g['num_samples']
    assert ds.get_feature_shape() == g['feature_shape'] and ds.get_label_shape() == g['label_shape']
    for i, gb in enumerate(g['batches']):
        assert ds.has_more_batches()
        m, l, s, n = ds.get_next_batch()
        np.testing.assert_array_equal(m, BATCHES['b%d_mfccs' % i])
        np.testing.assert_array_equal(l, BATCHES['b%d_labels' % i])
        assert str(m.dtype) == gb['mfccs_dtype'] and str(l.dtype) == gb['labels_dtype']
        assert [int(x) for x in s] == gb['seq_len'] and n == gb['labels_len']
        assert type(s[0]).__name__ == gb['seq_len_type'] and s[0].ndim == 0 and str(s[0].dtype) == gb['seq_len_dtype']
        assert type(n[0]).__name__ == gb['labels_len_type']
        assert ds.index == gb['index_after']
    assert not ds.has_more_batches()
    # tail batch = copies of the last file
    assert (BATCHES['b1_mfccs'][0] == BATCHES['b1_mfccs'][3]).all()
    ds.reset_epoch()
    assert ds.index == g['index_after_reset'] and ds.has_more_batches()


def test_dataset_rand_shift_matches_reference(toy_config):
    c = Config(toy_config, True)
    c.rand_shift = GOLD['dataset']['aug']['rand_shift']
    ds = DataSet(c.train_input, c)
    np.random.seed(GOLD['dataset']['aug']['seed'])
    m, l, s, n = ds.get_next_batch()
    np.testing.assert_array_equal(m, BATCHES['aug_mfccs'])
    assert [int(x) for x in s] == GOLD['dataset']['aug']['seq_len']


def test_dataset_refuses_code_in_pickles(tmp_path, toy_config):
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))
    p = tmp_path / 'evil.pkl'
    p.write_bytes(pickle.dumps(Evil()))
    scp = tmp_path / 'x.scp'
    scp.write_text('evil.pkl\n')
    c = Config(toy_config, True)
    ds = DataSet(str(scp), c)
    with pytest.raises(pickle.UnpicklingError):
        ds.get_next_batch()


def test_include_context_matches_survey_golden():
    """arange(12).reshape(4,3), context 1."""
    out = utils.include_context(np.arange(12).reshape(4, 3), 1, 3)
    assert out.shape == (4, 9)
    assert out[0].tolist() == [0, 0, 0, 0, 1, 2, 3, 4, 5]
    assert out[3].tolist() == [6, 7, 8, 9, 10, 11, 0, 0, 0]
    assert out[1].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8]


def test_sparse_tuple_from_matches_survey_golden():
    i, v, s = utils.sparse_tuple_from([[1, 2, 0], [3, 0, 0]], [2, 1])
    assert i.tolist() == [[0, 0], [0, 1], [1, 0]] and i.dtype == np.int64
    assert v.tolist() == [1, 2, 3] and v.dtype == np.int32
    assert s.tolist() == [2, 2] and s.dtype == np.int64


def test_prefetch_yields_the_same_batches_in_order(toy_config):
    c = Config(toy_config, True)
    a, b = DataSet(c.train_input, c), DataSet(c.train_input, c)
    got = list(b.prefetch(depth=2))
    want = []
    while a.has_more_batches():
        want.append(a.get_next_batch())
    assert len(got) == len(want) == 2 and not b.has_more_batches()
    for (m1, l1, s1, n1), (m2, l2, s2, n2) in zip(got, want):
        np.testing.assert_array_equal(m1, m2)
        np.testing.assert_array_equal(l1, l2)
        assert [int(x) for x in s1] == [int(x) for x in s2] and n1 == n2
    b.reset_epoch()
    assert len(list(b.prefetch())) == 2          # a fresh epoch can be prefetched again


def test_prefetch_surfaces_loader_errors(toy_config, tmp_path):
    c = Config(toy_config, True)
    scp = tmp_path / 'bad.scp'
    scp.write_text('does_not_exist.pkl\n')
    with pytest.raises(FileNotFoundError):
        list(DataSet(str(scp), c).prefetch())
