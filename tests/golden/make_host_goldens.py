"""Generates tests/golden/host_goldens.json + host_batches.npz + sample_set/ by RUNNING the
reference's own host modules (config.py, symbols.py, dataset.py, audiosample.py) in the dev
container.  Run once here (`python tests/golden/make_host_goldens.py`); the outputs are
committed because /root/reference does not exist on the GPU box.  Only data is written:
inputs (a synthetic .scp/.pkl set, a config we wrote) and the outputs the reference produced.
utils.py is not imported (it needs librosa / python_speech_features, absent here); its two
pure-NumPy helpers are pinned by the values recorded in SURVEY.md §8c(4)."""
import json
import os
import pickle
import shutil
import sys

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'sample_set')

CONFIG_TEXT = """[Parameters]
samplerate=8000
numcep=3
numcontext=1
label_context=0
batch_size=2
epochs=2
learningrate=0.005
model_dir=%(model_dir)s
start_step=0
report_step=2
num_gpus=2
punc_regex=[^a-z0-9 ]
sym_file=${MFCC Featurizer:output}/symbols
network=networks.bilstm_ctc_net.BiLstmCTCNet

[Train]
input=${MFCC Featurizer:output}/train.scp

[Test]
input=${MFCC Featurizer:output}/test.scp

[MFCC Featurizer]
input=unused.csv
output=%(out)s
"""


def main():
    sys.path.insert(0, REF)
    import audiosample
    import config as refconfig
    import dataset as refdataset
    import symbols as refsymbols

    if os.path.exists(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    gold = {}

    # ---- Symbols: insert order padding, a, b, _, blank
    sym = refsymbols.Symbols(0)
    sym.insert_padding()
    for s in 'ab_':
        sym.insert_sym(s)
    sym.insert_blank()
    sym_path = os.path.join(OUT, 'symbols')
    sym.write(sym_path)
    gold['symbols_file_text'] = open(sym_path).read()
    sym2 = refsymbols.Symbols(0, sym_path)
    gold['symbols_counter'] = sym2.counter
    gold['symbols_padding_id'] = sym2.get_padding_id()
    gold['symbols_sym_to_id'] = sym2.sym_to_id
    gold['symbols_convert'] = {'ids': [1, 3, 2, 4, 1], 'str': sym2.convert_to_str([1, 3, 2, 4, 1])}

    # ---- synthetic utterances: 5 train + 2 test, feature_size = (2*1+1)*3 = 9
    rs = np.random.RandomState(42)
    shapes = [(7, 3), (5, 2), (9, 4), (6, 1), (8, 3)]
    names = []
    for i, (T, L) in enumerate(shapes):
        mf = rs.randn(T, 9).astype(np.float32)
        lab = rs.randint(1, 4, size=L).astype(np.int32)
        a = audiosample.AudioSample('utt%d' % i, mf, lab, ''.join('ab_'[k - 1] for k in lab))
        with open(os.path.join(OUT, 'utt%d.pkl' % i), 'wb') as f:
            pickle.dump(a, f, 2)
        names.append('utt%d.pkl' % i)
    with open(os.path.join(OUT, 'train.scp'), 'w') as f:
        f.write('\n'.join(names) + '\n')
    with open(os.path.join(OUT, 'test.scp'), 'w') as f:
        f.write('\n'.join(names[:2]) + '\n')

    cfg_path = os.path.join(OUT, 'toy.config')
    with open(cfg_path, 'w') as f:
        f.write(CONFIG_TEXT % dict(model_dir='.model_toy', out=OUT))
    cfg = refconfig.Config(cfg_path, True)
    gold['config'] = {k: getattr(cfg, k) for k in
                      ['samplerate', 'numcep', 'numcontext', 'rand_shift', 'feature_size', 'batch_size', 'epochs',
                       'learningrate', 'model_dir', 'start_step', 'report_step', 'num_gpus', 'label_context',
                       'punc_regex', 'network', 'start_marker', 'end_marker']}
    gold['config']['sym_file_basename'] = os.path.basename(cfg.sym_file)
    gold['config']['train_input_basename'] = os.path.basename(cfg.train_input)
    gold['config']['test_input_basename'] = os.path.basename(cfg.test_input)
    gold['config']['symbols_counter'] = cfg.symbols.counter

    # ---- DataSet batches (global batch 4 over 5 files -> 2 batches, tail = copies of file 5)
    ds = refdataset.DataSet(cfg.train_input, cfg)
    gold['dataset'] = {'num_samples': ds.get_num_of_sample(), 'feature_shape': ds.get_feature_shape(),
                       'label_shape': ds.get_label_shape(), 'batches': []}
    arrays = {}
    i = 0
    while ds.has_more_batches():
        mf, lab, sl, ll = ds.get_next_batch()
        arrays['b%d_mfccs' % i] = mf
        arrays['b%d_labels' % i] = lab
        gold['dataset']['batches'].append({
            'mfccs_shape': list(mf.shape), 'mfccs_dtype': str(mf.dtype),
            'labels_shape': list(lab.shape), 'labels_dtype': str(lab.dtype),
            'seq_len': [int(x) for x in sl], 'seq_len_type': type(sl[0]).__name__, 'seq_len_dtype': str(sl[0].dtype),
            'seq_len_ndim': int(sl[0].ndim), 'labels_len': [int(x) for x in ll], 'labels_len_type': type(ll[0]).__name__,
            'index_after': ds.index})
        i += 1
    ds.reset_epoch()
    gold['dataset']['index_after_reset'] = ds.index

    # ---- rand_shift augmentation with a fixed numpy seed
    cfg.rand_shift = 2
    ds2 = refdataset.DataSet(cfg.train_input, cfg)
    np.random.seed(123)
    mf, lab, sl, ll = ds2.get_next_batch()
    arrays['aug_mfccs'] = mf
    gold['dataset']['aug'] = {'seed': 123, 'rand_shift': 2, 'seq_len': [int(x) for x in sl], 'mfccs_shape': list(mf.shape)}

    np.savez(os.path.join(HERE, 'host_batches.npz'), **arrays)
    with open(os.path.join(HERE, 'host_goldens.json'), 'w') as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    for junk in ('.model_toy',):
        if os.path.exists(junk):
            shutil.rmtree(junk)
    print('wrote', os.path.join(HERE, 'host_goldens.json'))


if __name__ == '__main__':
    main()
