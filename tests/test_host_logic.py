"""Host logic around the hot path that needs no GPU: the training loop's cadence and log lines
(reference: train.py:14-47) with a stub network, greedy-collapse / LER helper in the C library (host code)."""
import logging
import os

import numpy as np
import pytest

from neuralasr_amd import train as train_mod
from neuralasr_amd.config import Config
from neuralasr_amd.dataset import DataSet
from oracle import nasr_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLES = os.path.join(HERE, 'golden', 'sample_set')


class StubNet:
    def __init__(self, config, fortraining=False):
        self.global_step = config.start_step
        self.saved, self.validated, self.batches = [], 0, []

    def train(self, mfccs, labels, seq_len, labels_len):
        self.global_step += 1
        self.batches.append(mfccs.shape)
        return np.float32(2.0 * self.global_step), np.float32(0.5)

    def validate(self, *a):
        self.validated += 1
        return [np.float32(1.25), np.float32(0.75)]

    def save_checkpoint(self):
        self.saved.append(self.global_step)


def test_train_loop_cadence_and_log_formats(tmp_path, caplog):
    text = open(os.path.join(SAMPLES, 'toy.config')).read().splitlines()
    text = ['output=' + SAMPLES if l.startswith('output=') else l for l in text]
    text = ['model_dir=' + str(tmp_path / 'm') if l.startswith('model_dir=') else l for l in text]
    cfgp = tmp_path / 'c.config'
    cfgp.write_text('\n'.join(text) + '\n')
    cfg = Config(str(cfgp), True)
    holder = {}

    def fake_load(fortraining=False):
        holder['net'] = StubNet(cfg, fortraining)
        return holder['net']
    cfg.load_network = fake_load
    train = DataSet(cfg.train_input, cfg)
    valid = DataSet(cfg.test_input, cfg)
    with caplog.at_level(logging.INFO, logger='NeuralASR'):
        train_mod.train_model(train, valid, cfg)
    net = holder['net']
    # 2 epochs x 2 batches (5 files, global batch 4), report_step 2
    assert net.global_step == 4 and net.saved == [2, 4] and net.validated == 2
    assert net.batches == [(4, 9, 9), (4, 8, 9)] * 2
    msgs = [r.getMessage() for r in caplog.records]
    steps = [m for m in msgs if m.startswith('Step: ')]
    assert steps[0].startswith('Step: 0002, cost = 3.0000, ler = 0.5000, time = ')       # mean of 2.0, 4.0
    assert steps[1].startswith('Step: 0004, cost = 7.0000, ler = 0.5000, time = ')       # window reset
    assert 'Valid: cost = 1.2500, ler = 0.7500' in msgs
    assert msgs[-1] == 'Finished training!!!'
    t1 = float(steps[0].rsplit('= ', 1)[1])
    t2 = float(steps[1].rsplit('= ', 1)[1])
    assert t2 >= t1                                             # train_time_sec is cumulative


def test_label_error_rate_host_function_matches_oracle():
    from neuralasr_amd import _lib
    import ctypes
    lib = _lib.load()
    rs = np.random.RandomState(3)
    B, Lmax = 6, 7
    labels = rs.randint(0, 4, size=(B, Lmax)).astype(np.int32)
    label_len = np.array([7, 3, 0, 5, 1, 0], np.int32)
    hyps = [rs.randint(0, 4, size=n).tolist() for n in (5, 3, 0, 9, 0, 0)]
    hyps[1] = labels[1, :3].tolist()
    ids = np.zeros((B, 9), np.int32)
    lens = np.zeros(B, np.int32)
    for b, h in enumerate(hyps):
        ids[b, :len(h)] = h
        lens[b] = len(h)
    out = ctypes.c_float()
    ip = ctypes.POINTER(ctypes.c_int32)
    rc = lib.nasr_label_error_rate(ids.ctypes.data_as(ip), lens.ctypes.data_as(ip), 9, labels.ctypes.data_as(ip),
                                   label_len.ctypes.data_as(ip), Lmax, B, ctypes.byref(out))
    assert rc == 0
    assert out.value == pytest.approx(O.label_error_rate(hyps, labels, label_len), rel=1e-6)
    # empty truth + non-empty hypothesis -> inf, as tf.edit_distance(normalize=True) does
    lens[2] = 2
    lib.nasr_label_error_rate(ids.ctypes.data_as(ip), lens.ctypes.data_as(ip), 9, labels.ctypes.data_as(ip),
                              label_len.ctypes.data_as(ip), Lmax, B, ctypes.byref(out))
    assert np.isinf(out.value)
