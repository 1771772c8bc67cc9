"""Host logic around the hot path that needs no GPU: the training loop's cadence and log lines
(reference: train.py:14-47) with a stub network, greedy-collapse / LER helper in the C library (host code)."""
import ctypes
import logging
import os

import numpy as np
import pytest

from neuralasr_amd import train as train_mod
from neuralasr_amd.config import Config
from neuralasr_amd.dataset import DataSet
from oracle import nasr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def test_deepspeech_class_mirrors_the_reference_locals():
    """networks/deepspeech.py:15-26: widths, clip, stddev, seed; dropout of layers 1, 2, 3 and 5 (the two cell entries
    of the reference's list are 0.0 = no DropoutWrapper effect)."""
    from neuralasr_amd.networks.deepspeech import DeepSpeech
    assert DeepSpeech.n_hidden == 2048 and DeepSpeech.n_cell_dim == 2048
    assert DeepSpeech.pre_widths() == (2048, 2048, 4096)              # n_hidden_1, n_hidden_2, n_hidden_3 = 2*n_cell_dim
    assert DeepSpeech.relu_clip == 20.0 and DeepSpeech.stddev == 0.046875 and DeepSpeech.random_seed == 4567
    assert DeepSpeech.dropout == (0.05, 0.05, 0.05, 0.05)
    assert DeepSpeech.bidirectional and DeepSpeech.merge == 'concat' and DeepSpeech.num_layers == 1
    # initial values per the reference's initialisers, on a toy tensor list (name, offset, rows, cols)
    tensors = [('b1', 0, 8, 1), ('h1', 8, 6, 8), ('h2', 56, 8, 8), ('l0/fw/kernel', 120, 12, 16), ('l0/fw/bias', 312, 16, 1),
               ('b6', 328, 5, 1), ('h6', 333, 8, 5)]
    net = DeepSpeech.__new__(DeepSpeech)
    p = net.initial_params(tensors, seed=3)
    assert p.dtype == np.float32 and p.size == 373
    assert np.all(p[312:328] == 0)                                    # cell bias: zeros (TF default)
    lim = np.sqrt(6.0 / (12 + 16))
    assert np.abs(p[120:312]).max() <= lim                            # cell kernel: glorot-uniform (TF default)
    assert 0 < np.abs(p[0:8]).max() < 0.3                             # b1 ~ N(0, stddev)


def test_engine_packs_the_dense_stage_fields():
    """ctypes layout of the DeepSpeech fields of nasr_model_cfg (include/nasr.h) as Engine fills them."""
    from neuralasr_amd import _lib
    cfg = _lib.ModelCfg(546, 2048, 1, 1, 2, 29, 1.0, 1e-4, 0.9, 0.999, 1e-8, 3, (ctypes.c_int32 * 3)(2048, 2048, 4096), 2048,
                        20.0, (ctypes.c_float * 4)(0.05, 0.05, 0.05, 0.05))
    raw = bytes(cfg)
    ints = np.frombuffer(raw[44:44 + 20], np.int32)                   # after 6 int32 + 5 float
    assert ints.tolist() == [3, 2048, 2048, 4096, 2048]
    fl = np.frombuffer(raw[64:64 + 20], np.float32)
    assert fl.tolist() == pytest.approx([20.0, 0.05, 0.05, 0.05, 0.05])


def test_bench_algorithmic_bytes_cover_the_dense_stages():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s3, _ = bench.workload_spec('bilstm3x500')
    A, W, R = bench.algorithmic_bytes(s3, 16, 500)
    assert (A, W, R) == (2246656000, 40 * s3.param_count(), 24000000000)          # 2 247 MB, 649 MB, 24 000 MB
    sd, name = bench.workload_spec('deepspeech')
    assert sd.pre == (2048, 2048, 4096) and sd.post == 2048 and 'deepspeech' in name
    Ad, Wd, Rd = bench.algorithmic_bytes(sd, 32, 500)
    assert Ad > A and Wd == 40 * sd.param_count() and Rd == 2 * 500 * 2 * 4 * 2048 * 2048 * 4


def test_collective_without_process_group_offers_no_bucketed_reducer():
    """Single process (no torch.distributed group): the bucketed exchange is not offered, the plain path is used."""
    from neuralasr_amd.parallel import Collective

    class FakeEngine:
        def grad_buckets(self):
            return [(32, 10), (0, 32)]
    assert Collective().bucketed(FakeEngine(), None) is None


def test_context_structure_check_refuses_every_rand_shift_crop():
    """rand_shift's roll-and-crop (dataset.py:23-31) breaks include_context's window structure only in the
    first / last numcontext frames of an utterance (real neighbours where the pad value was, a pad read from a real
    sample): a sampled check passed 98 % of such batches.  The edge frames are now checked exactly."""
    from neuralasr_amd.engine import Engine
    from neuralasr_amd.utils import include_context
    ctx, ncep, B = 10, 26, 16
    rs = np.random.RandomState(3)
    refused = 0
    for trial in range(40):
        T = 120
        feats = np.zeros((B, T, (2 * ctx + 1) * ncep), np.float32)
        seq = np.zeros(B, np.int32)
        shifted = trial % 2 == 1
        for b in range(B):
            n = rs.randint(60, T + 1)
            st = include_context(rs.randn(n, ncep).astype(np.float32), ctx, ncep)
            st = ((st - st.mean()) / st.std()).astype(np.float32)
            if shifted and b == trial % B:                      # one utterance of the batch went through augment_mfcc
                r = [-2, -1, 1, 2][(trial // 2) % 4]
                st = np.roll(st, r, axis=0)
                st = st[r:] if r > 0 else st[:r]
            seq[b] = st.shape[0]
            feats[b, :seq[b]] = st
        ok = Engine.context_structure_ok(feats, seq, ctx, ncep)
        assert ok == (not shifted), (trial, ok)
        refused += not ok
    assert refused == 20
    assert not Engine.context_structure_ok(feats[:, :, :-1], seq, ctx, ncep)      # wrong width


def test_prefetch_worker_stops_when_the_consumer_walks_away(tmp_path):
    """An abandoned prefetch generator (a train step raised) must release its loader thread."""
    import threading
    import time
    from neuralasr_amd.config import Config
    from neuralasr_amd.dataset import DataSet
    samples = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sample_set')
    lines = open(os.path.join(samples, 'toy.config')).read().replace('batch_size=2', 'batch_size=1')
    cfgp = tmp_path / 'toy.config'
    cfgp.write_text('\n'.join(('output=' + samples) if ln.startswith('output=') else ln for ln in lines.splitlines()) + '\n')
    cfg = Config(str(cfgp), True)
    ds = DataSet(cfg.train_input, cfg)
    gen = ds.prefetch(depth=1)
    next(gen)
    gen.close()                                   # consumer stops after one batch: the queue is full, the worker blocked
    deadline = time.time() + 5.0
    while time.time() < deadline and any(t.name == 'nasr-prefetch' and t.is_alive() for t in threading.enumerate()):
        time.sleep(0.05)
    assert not any(t.name == 'nasr-prefetch' and t.is_alive() for t in threading.enumerate())


def test_nested_prefetch_generators_restore_the_switch_interval(tmp_path):
    """DataSet.prefetch lowers sys.setswitchinterval while a loader thread runs.  Two generators alive at once (train +
    validation) closed in any order - or one of them abandoned - must leave the interval the process started with."""
    import gc
    import sys
    samples = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sample_set')
    lines = open(os.path.join(samples, 'toy.config')).read().replace('batch_size=2', 'batch_size=1')
    cfgp = tmp_path / 'toy.config'
    cfgp.write_text('\n'.join(('output=' + samples) if ln.startswith('output=') else ln for ln in lines.splitlines()) + '\n')
    cfg = Config(str(cfgp), True)
    start = sys.getswitchinterval()
    a, b = DataSet(cfg.train_input, cfg).prefetch(depth=1), DataSet(cfg.test_input, cfg).prefetch(depth=1)
    next(a)
    assert sys.getswitchinterval() <= min(start, 2e-4)
    next(b)
    a.close()                                     # the first one in leaves first: the override must outlive it
    assert sys.getswitchinterval() <= min(start, 2e-4)
    del b                                         # abandoned, never closed
    gc.collect()
    assert sys.getswitchinterval() == start


def test_train_loop_leaves_void_steps_out_of_its_means(tmp_path, caplog):
    """A step whose forward pass was void reports NaN on every rank (HipNetwork repeats it one call later): the window's
    means are taken over the steps that count, the log never shows `cost = nan`."""
    text = open(os.path.join(SAMPLES, 'toy.config')).read().splitlines()
    text = ['output=' + SAMPLES if l.startswith('output=') else l for l in text]
    text = ['model_dir=' + str(tmp_path / 'm') if l.startswith('model_dir=') else l for l in text]
    cfgp = tmp_path / 'c.config'
    cfgp.write_text('\n'.join(text) + '\n')
    cfg = Config(str(cfgp), True)

    class VoidNet(StubNet):
        def train(self, *a):
            loss, ler = StubNet.train(self, *a)
            return (np.float32('nan'), np.float32('nan')) if self.global_step == 1 else (loss, ler)
    cfg.load_network = lambda fortraining=False: VoidNet(cfg, fortraining)
    with caplog.at_level(logging.INFO, logger='NeuralASR'):
        train_mod.train_model(DataSet(cfg.train_input, cfg), None, cfg)
    steps = [r.getMessage() for r in caplog.records if r.getMessage().startswith('Step: ')]
    assert steps[0].startswith('Step: 0002, cost = 4.0000, ler = 0.5000')       # step 1 (NaN) left out: mean of {4.0}
    assert steps[1].startswith('Step: 0004, cost = 7.0000, ler = 0.5000')
