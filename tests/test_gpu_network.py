"""GPU: the `Network` plugin surface end to end (config -> load_network -> train / validate / evaluate /
decode / checkpoints) against the oracle's restatement of the reference's tower arithmetic."""
import glob
import os

import numpy as np
import pytest

from neuralasr_amd.config import Config
from neuralasr_amd.dataset import DataSet
from oracle import nasr_oracle as O

pytestmark = pytest.mark.gpu
needs_persistent = pytest.mark.skipif(os.environ.get('NASR_PERSIST', '1')[:1] == '0',
                                      reason='asserts the persistent recurrence; NASR_PERSIST=0 forces the per-step kernels')
HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLES = os.path.join(HERE, 'golden', 'sample_set')


def make_config(tmp_path, **over):
    lines = open(os.path.join(SAMPLES, 'toy.config')).read().splitlines()
    over = dict({'output': SAMPLES, 'model_dir': str(tmp_path / 'model')}, **over)
    out = []
    for ln in lines:
        key = ln.split('=')[0]
        out.append('%s=%s' % (key, over[key]) if key in over and '=' in ln else ln)
    p = tmp_path / 'toy.config'
    p.write_text('\n'.join(out) + '\n')
    return str(p)


def spec_of(net, cfg):
    return O.ModelSpec(cfg.feature_size, net.num_hidden, net.num_layers, net.bidirectional, net.merge,
                       cfg.symbols.counter)


def test_load_network_by_reference_name_and_train_two_towers(tmp_path):
    cfg = Config(make_config(tmp_path), True)
    assert cfg.network == 'networks.bilstm_ctc_net.BiLstmCTCNet' and cfg.num_gpus == 2 and cfg.batch_size == 4
    net = cfg.load_network(fortraining=True)
    assert type(net).__name__ == 'BiLstmCTCNet' and net.engine.backend == 'hip-gfx950'
    assert os.path.exists(os.path.join(cfg.model_dir, 'toy.config'))            # config + symbols copied
    assert os.path.exists(os.path.join(cfg.model_dir, 'symbols'))
    spec = spec_of(net, cfg)
    params = [p.astype(np.float32).astype(np.float64) for p in O.unflatten(spec, net.engine.get_params())]
    ds = DataSet(cfg.train_input, cfg)
    mfccs, labels, seq_len, labels_len = ds.get_next_batch()
    # oracle: two towers on contiguous halves, D3 map per tower, mean of tower means, TF Adam
    loss_o, grads_o = O.data_parallel_loss_and_grads(spec, params, mfccs, [int(s) for s in seq_len], labels,
                                                     labels_len, 2)
    lers = []
    for sl in O.shard_slices(4, 2):
        lg, _ = O.network_forward(spec, params, mfccs[sl], [int(s) for s in seq_len[sl]])
        # the mean_ler of a training step is the reference's: ctc_beam_search_decoder defaults (tfnetwork.py:61-70,188-189)
        lens = [int(s) for s in seq_len[sl]]
        hy = [O.ctc_beam_search(lg[:lens[b], b], 100, True)[0] for b in range(len(lens))]
        lers.append(O.label_error_rate(hy, labels[sl], labels_len[sl]))
    newp, _, _ = O.adam_tf(params, grads_o, [0 * p for p in params], [0 * p for p in params], 1, cfg.learningrate)
    loss, ler = net.train(mfccs, labels, seq_len, labels_len)
    assert isinstance(loss, np.float32) and isinstance(ler, np.float32)
    assert net.global_step == 1
    assert float(loss) == pytest.approx(loss_o, rel=2e-5)
    assert float(ler) == pytest.approx(np.mean(lers), abs=1e-6)
    got = net.engine.get_params()
    assert np.abs(got - O.flatten(newp)).max() < 0.02 * cfg.learningrate + 1e-6
    v = net.validate(mfccs, labels, seq_len, labels_len)
    assert isinstance(v, list) and len(v) == 2


def test_validate_and_train_with_a_global_batch_above_the_per_gpu_limit(tmp_path):
    """Two configs the reference ships have batch_size x num_gpus > 64 (16000sr_26mfcc_full: 32 x 4, 8000sr_13mfcc:
    20 x 4).  A training network evaluates per tower (make_parallel, tfnetwork.py:88-137): validate() must split the
    global batch exactly as train() does - one engine call per tf.split shard, mean of the shard means - instead of
    handing 80 utterances to one handle (per-GPU limit 64)."""
    cfg = Config(make_config(tmp_path, network='networks.lstm_ctc_net.SmallLstmCTCNet', batch_size='20', num_gpus='4'), True)
    assert cfg.batch_size == 80
    net = cfg.load_network(fortraining=True)
    net.decoder = 'greedy'
    spec = spec_of(net, cfg)
    params = [p.astype(np.float32).astype(np.float64) for p in O.unflatten(spec, net.engine.get_params())]
    mfccs, labels, seq_len, labels_len = DataSet(cfg.train_input, cfg).get_next_batch()
    assert mfccs.shape[0] == 80
    seq = [int(s) for s in seq_len]
    losses, lers = [], []
    for sl in O.shard_slices(80, 4):
        lo, _, _, _ = O.network_loss_and_grads(spec, params, mfccs[sl], seq[sl], labels[sl], labels_len[sl])
        lg, _ = O.network_forward(spec, params, mfccs[sl], seq[sl])
        lers.append(O.label_error_rate(O.greedy_decode(lg, seq[sl]), labels[sl], labels_len[sl]))
        losses.append(lo)
    vloss, vler = net.validate(mfccs, labels, seq_len, labels_len)
    assert float(vloss) == pytest.approx(np.mean(losses), rel=2e-5)
    assert float(vler) == pytest.approx(np.mean(lers), abs=1e-6)
    loss, _ = net.train(mfccs, labels, seq_len, labels_len)
    assert float(loss) == pytest.approx(np.mean(losses), rel=2e-5)
    with pytest.raises(Exception, match=r'per-GPU batch must be in \[1,64\]'):
        net.engine.loss(mfccs, seq_len, labels, labels_len)          # what the unsplit call would have hit


def test_checkpoint_cadence_resume_and_wipe(tmp_path):
    cfg = Config(make_config(tmp_path, num_gpus=1, batch_size=4), True)
    net = cfg.load_network(fortraining=True)
    ds = DataSet(cfg.train_input, cfg)
    b = ds.get_next_batch()
    for _ in range(7):
        net.train(*[b[0], b[1], b[2], b[3]])
        net.save_checkpoint()
    files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(cfg.model_dir, 'model-*.npz')))
    assert files == ['model-%d.npz' % s for s in (3, 4, 5, 6, 7)]                 # Saver keeps 5
    want = net.engine.get_params()
    m, v, step = net.engine.get_adam_state()
    assert step == 7
    # resume: start_step > 0 restores the latest checkpoint (tfnetwork.py:142-147)
    cfg2 = Config(make_config(tmp_path, num_gpus=1, batch_size=4, start_step=7), True)
    net2 = cfg2.load_network(fortraining=True)
    assert net2.global_step == 7
    np.testing.assert_array_equal(net2.engine.get_params(), want)
    m2, v2, step2 = net2.engine.get_adam_state()
    assert step2 == 7
    np.testing.assert_array_equal(m2, m)
    l_a, _ = net.train(*b)
    l_b, _ = net2.train(*b)
    assert l_a == l_b                                                              # bitwise same continuation
    # inference always restores (tfnetwork.py:46-51)
    cfg3 = Config(make_config(tmp_path, num_gpus=1, batch_size=1), True)
    net3 = cfg3.load_network(fortraining=False)
    np.testing.assert_array_equal(net3.engine.get_params(), want)
    # start_step == 0 wipes model_dir (tfnetwork.py:148-151)
    cfg4 = Config(make_config(tmp_path, num_gpus=1, batch_size=4), True)
    cfg4.load_network(fortraining=True)
    assert glob.glob(os.path.join(cfg4.model_dir, 'model-*.npz')) == []


def test_evaluate_and_decode_at_batch_one(tmp_path):
    cfg = Config(make_config(tmp_path, num_gpus=1, batch_size=1), True)
    net = cfg.load_network(fortraining=True)
    spec = spec_of(net, cfg)
    params = [p.astype(np.float64) for p in O.unflatten(spec, net.engine.get_params())]
    ds = DataSet(cfg.test_input, cfg)
    mfccs, labels, seq_len, labels_len = ds.get_next_batch()
    sl = [int(s) for s in seq_len]
    lo, nll, _, logits = O.network_loss_and_grads(spec, params, mfccs, sl, labels, labels_len)
    # default decoder = the reference's: beam width 100, merge_repeated (tfnetwork.py:61-64)
    ids, loss, ler = net.evaluate(mfccs, labels, seq_len, labels_len)
    hy_beam = [O.ctc_beam_search(logits[:sl[0], 0], 100, True)[0]]
    assert ids.dtype == np.int64 and ids.tolist() == hy_beam[0]
    assert float(loss) == pytest.approx(lo, rel=2e-5)
    assert float(ler) == pytest.approx(O.label_error_rate(hy_beam, labels, labels_len), abs=1e-6)
    assert net.decode(mfccs, seq_len).tolist() == hy_beam[0]
    # greedy variant (the decoder named in the comment at tfnetwork.py:62-63)
    net.decoder = 'greedy'
    hy = O.greedy_decode(logits, sl)
    ids, loss, ler = net.evaluate(mfccs, labels, seq_len, labels_len)
    assert ids.tolist() == hy[0]
    assert float(ler) == pytest.approx(O.label_error_rate(hy, labels, labels_len), abs=1e-6)
    assert net.decode(mfccs, seq_len).tolist() == hy[0]
    assert isinstance(cfg.symbols.convert_to_str(ids), str)


def test_train_model_loop_end_to_end(tmp_path, caplog):
    import logging
    from neuralasr_amd import train as train_mod
    cfg = Config(make_config(tmp_path, network='networks.lstm_ctc_net.SmallLstmCTCNet', epochs=3), True)
    train = DataSet(cfg.train_input, cfg)
    valid = DataSet(cfg.test_input, Config(make_config(tmp_path, network='networks.lstm_ctc_net.SmallLstmCTCNet'), True))
    with caplog.at_level(logging.INFO, logger='NeuralASR'):
        net = train_mod.train_model(train, valid, cfg)
    assert net.global_step == 6
    steps = [r.getMessage() for r in caplog.records if r.getMessage().startswith('Step: ')]
    assert len(steps) == 3
    costs = [float(m.split('cost = ')[1].split(',')[0]) for m in steps]
    assert costs[-1] < costs[0]                                   # it learns the toy set
    assert sorted(os.path.basename(f) for f in glob.glob(os.path.join(cfg.model_dir, 'model-*.npz'))) == \
        ['model-2.npz', 'model-4.npz', 'model-6.npz']


def test_staged_input_pipeline_trains_bit_equal_to_the_synchronous_path(tmp_path):
    """train_model with the loader thread staging every next batch (pinned memory + copy stream, HipNetwork.stage_batch)
    against the same loop with synchronous uploads: identical parameters after 3 epochs, and the staged path was the one
    that ran."""
    from neuralasr_amd import train as train_mod
    from neuralasr_amd.networks import hipnetwork
    nets = []
    for k, prefetch in enumerate((2, 0)):
        cfg = Config(make_config(tmp_path, network='networks.lstm_ctc_net.SmallLstmCTCNet', epochs=3, num_gpus='1',
                                 model_dir=str(tmp_path / ('m%d' % k))), True)
        commits = []
        orig = hipnetwork.Engine.commit_batch
        hipnetwork.Engine.commit_batch = lambda self, t, _o=orig, _c=commits: (_c.append(t), _o(self, t))[1]
        try:
            nets.append((train_mod.train_model(DataSet(cfg.train_input, cfg), None, cfg, prefetch=prefetch), len(commits)))
        finally:
            hipnetwork.Engine.commit_batch = orig
    (a, na), (b, nb) = nets
    # the loader runs up to three batches ahead and at most two may be staged: the odd one is uploaded synchronously
    assert a.global_step == b.global_step == 9 and 5 <= na <= 9 and nb == 0
    np.testing.assert_array_equal(a.engine.get_params(), b.engine.get_params())
    m1, v1, s1 = a.engine.get_adam_state()
    m2, v2, s2 = b.engine.get_adam_state()
    assert s1 == s2 == 9
    np.testing.assert_array_equal(m1, m2)


def test_rccl_path_at_world_one_orders_with_the_engine_stream(tmp_path):
    """The data-parallel step (compute_grads -> all_reduce on the aliased device buffer -> apply_adam(1/n)) under
    a real NCCL(RCCL) process group of one rank must equal the plain step bit for bit, and the torch tensor
    must ALIAS the engine's gradient buffer."""
    import torch
    import torch.distributed as dist
    from neuralasr_amd.engine import Engine
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        spec = O.ModelSpec(20, 48, 2, True, 'concat', 11)
        feats, seq_len, labels, label_len = O.synth_batch(spec, 8, 21, seed=3, var_len=True, Lmin=1, Lmax=5)
        p0 = O.flatten(O.init_params(spec, seed=8)).astype(np.float32)
        ref = Engine(20, 48, 2, True, 'concat', 11, learning_rate=1e-3)
        ref.set_params(p0)
        for _ in range(3):
            ref.train_step(feats, seq_len, labels, label_len)
        ts = torch.cuda.Stream()
        torch.cuda.set_stream(ts)
        with pytest.raises(ValueError):
            Engine(20, 48, 2, True, 'concat', 11, stream=0)
        e = Engine(20, 48, 2, True, 'concat', 11, learning_rate=1e-3, stream=ts.cuda_stream)
        e.set_params(p0)
        gt = e.grad_tensor()
        assert gt.numel() == e.grad_device_ptr()[1] and gt.data_ptr() == e.grad_device_ptr()[0]
        for _ in range(3):
            e.upload_batch(feats, seq_len, labels, label_len)
            e.compute_grads()
            dist.all_reduce(gt, op=dist.ReduceOp.SUM)
            e.apply_adam(1.0)
        np.testing.assert_array_equal(e.get_params(), ref.get_params())
        # the same three steps with the exchange cut into per-layer buckets on a side stream (include/nasr.h,
        # "Overlapping the exchange"): buckets tile the buffer, the fault word leads it, parameters come out identical
        from neuralasr_amd.parallel import BucketedAllReduce, Collective
        b = Engine(20, 48, 2, True, 'concat', 11, learning_rate=1e-3, stream=ts.cuda_stream)
        b.set_params(p0)
        red = Collective().bucketed(b, b.grad_tensor())
        assert isinstance(red, BucketedAllReduce)
        buckets = b.grad_buckets()
        assert len(buckets) == 2 and buckets[-1][0] == 0 and buckets[0][0] == buckets[-1][1]
        assert sum(c for _, c in buckets) == b.grad_device_ptr()[1]
        for _ in range(3):
            b.upload_batch(feats, seq_len, labels, label_len)
            b.compute_grads()
            red.all_reduce()
            b.apply_adam(1.0)
        assert not b.step_void()
        np.testing.assert_array_equal(b.get_params(), ref.get_params())
        # a side-stream reader that waited for bucket 0 sees the finished top-layer gradients of THIS compute_grads
        b.upload_batch(feats, seq_len, labels, label_len)
        b.compute_grads()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            b.bucket_wait(0, side.cuda_stream)
            early = red.views[0].clone()
        torch.cuda.synchronize()
        assert torch.equal(early, red.views[0]) and float(early.abs().sum()) > 0
        b.close()
        # aliasing: scaling the torch view scales what the engine hands back
        e.upload_batch(feats, seq_len, labels, label_len)
        e.compute_grads()
        g1 = e.get_grads()
        gt.mul_(2.0)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(e.get_grads(), 2 * g1)
        e.close()
        ref.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
        dist.destroy_process_group()


@needs_persistent
def test_deepspeech_plugin_trains_and_validates(tmp_path, monkeypatch):
    """`network=networks.deepspeech.DeepSpeech` resolves to the HIP class (reference: networks/deepspeech.py); run at
    reduced widths so the toy set trains in a blink: variable names / shapes in the reference's creation order, initial
    values per its initialisers, loss falls, validate() and decode() work, checkpoints round-trip."""
    from neuralasr_amd.networks import deepspeech as ds_mod
    monkeypatch.setattr(ds_mod.DeepSpeech, 'n_hidden', 48)
    monkeypatch.setattr(ds_mod.DeepSpeech, 'n_cell_dim', 40)
    cfg = Config(make_config(tmp_path, network='networks.deepspeech.DeepSpeech', num_gpus='1', learningrate='0.002'), True)
    net = cfg.load_network(fortraining=True)
    assert type(net).__name__ == 'DeepSpeech' and net.engine.recurrence_mode == 'persistent'
    names = [n for n, _, _, _ in net.engine.tensors()]
    assert names == ['b1', 'h1', 'b2', 'h2', 'b3', 'h3', 'l0/fw/kernel', 'l0/fw/bias', 'l0/bw/kernel', 'l0/bw/bias',
                     'b5', 'h5', 'b6', 'h6']
    shapes = {n: (r, c) for n, _, r, c in net.engine.tensors()}
    F, C = cfg.feature_size, cfg.symbols.counter
    assert shapes['h1'] == (F, 48) and shapes['h3'] == (48, 80) and shapes['l0/fw/kernel'] == (80 + 40, 160)
    assert shapes['h5'] == (80, 48) and shapes['h6'] == (48, C)
    p0 = net.engine.get_params()
    off = {n: o for n, o, _, _ in net.engine.tensors()}
    assert abs(p0[off['h2']:off['h2'] + 48 * 48].std() - 0.046875) < 0.01          # N(0, stddev) (deepspeech.py:25,56)
    assert np.all(p0[off['l0/fw/bias']:off['l0/fw/bias'] + 160] == 0)
    ds = DataSet(cfg.train_input, cfg)
    mfccs, labels, seq_len, labels_len = ds.get_next_batch()
    losses = [float(net.train(mfccs, labels, seq_len, labels_len)[0]) for _ in range(8)]
    assert np.isfinite(losses).all() and min(losses[-3:]) < losses[0]
    assert net.global_step == 8 and net.engine.dropout_state() == (4567, 8)
    vl = net.validate(mfccs, labels, seq_len, labels_len)
    assert np.isfinite(vl[0])
    net.save_checkpoint()
    assert glob.glob(os.path.join(cfg.model_dir, 'model-8.npz'))
    ids = net.decode(mfccs[:1], seq_len[:1])
    assert ids.ndim == 1


@needs_persistent
def test_train_repeats_a_void_step(tmp_path, monkeypatch, caplog):
    """HipNetwork.train with a fault injected into the persistent recurrence (NASR_PERSIST_FAULT): the step is void,
    train() repeats it on the per-step kernels and returns what an undisturbed step returns."""
    import logging
    cfg = Config(make_config(tmp_path, num_gpus='1'), True)
    net = cfg.load_network(fortraining=True)
    ref = Config(make_config(tmp_path, num_gpus='1', model_dir=str(tmp_path / 'model2')), True).load_network(fortraining=True)
    ref.engine.set_params(net.engine.get_params())
    assert net.engine.recurrence_mode == 'persistent'
    ds = DataSet(cfg.train_input, cfg)
    mfccs, labels, seq_len, labels_len = ds.get_next_batch()
    want = ref.train(mfccs, labels, seq_len, labels_len)
    monkeypatch.setenv('NASR_PERSIST_FAULT', '2')
    with caplog.at_level(logging.WARNING):
        got = net.train(mfccs, labels, seq_len, labels_len)
    monkeypatch.delenv('NASR_PERSIST_FAULT')
    assert net.engine.recurrence_mode == 'per-step' and net.global_step == 1
    assert float(got[0]) == pytest.approx(float(want[0]), rel=2e-6) and float(got[1]) == pytest.approx(float(want[1]), abs=1e-6)
    np.testing.assert_allclose(net.engine.get_params(), ref.engine.get_params(), rtol=0, atol=5e-5)
    assert net.engine.get_adam_state()[2] == 1


@needs_persistent
def test_overfits_a_fixed_batch_to_zero_label_error():
    """End to end through the HIP path only: a small BiLSTM-CTC net trained on one fixed batch learns it by heart - the
    loss falls by orders of magnitude and the greedy and beam decodes reproduce the labels exactly (persistent
    recurrence, tiled-plane GEMMs, CTC, Adam, decoders all in the loop)."""
    from neuralasr_amd.engine import Engine
    spec = O.ModelSpec(20, 64, 1, True, 'concat', 12)
    B, T = 6, 40
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=17, var_len=True, Lmin=3, Lmax=6)
    e = Engine(spec.feature_size, spec.hidden, 1, True, 'concat', spec.num_classes, learning_rate=1e-2)
    assert e.recurrence_mode == 'persistent'
    e.set_params(O.flatten(O.init_params(spec, seed=4)).astype(np.float32))
    first = e.train_step(feats, seq_len, labels, label_len)
    for _ in range(400):
        last = e.train_step(feats, seq_len, labels, label_len)
    assert np.isfinite(last) and last < 0.02 * first
    truth = [labels[b, :label_len[b]].tolist() for b in range(B)]
    assert e.greedy_decode(feats, seq_len) == truth
    logits = e.forward(feats, seq_len)
    # ctc_beam_search_decoder's default merge_repeated=True also merges equal labels that a blank separated (SURVEY.md
    # Appendix A.6): the reference's decoder cannot emit "6 6"
    merged = [[v for i, v in enumerate(t) if i == 0 or v != t[i - 1]] for t in truth]
    assert e.beam_search(logits, seq_len, 100, merge_repeated=True)[0] == merged
    assert e.beam_search(logits, seq_len, 100, merge_repeated=False)[0] == truth
    assert e.label_error_rate(truth, labels, label_len) == 0.0
    e.close()


def test_in_library_rccl_exchange_at_world_one():
    """nasr_comm_* (include/nasr.h): a communicator of one rank bound through dlopen'ed librccl; the data-parallel step
    compute_grads -> comm_allreduce_grads (one ncclAllReduce per bucket on the communication stream, behind the bucket
    events) -> apply_adam(1/n) leaves the same parameters as the plain step, bit for bit; comm_mean of one rank is the
    identity; a second init is refused."""
    from neuralasr_amd import _lib
    from neuralasr_amd.engine import Engine
    spec = O.ModelSpec(20, 48, 3, True, 'concat', 9)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 6, 25, seed=3, var_len=True, Lmin=1, Lmax=4)
    p0 = O.flatten(O.init_params(spec, seed=4))
    a = Engine(20, 48, 3, True, 'concat', 9, learning_rate=1e-2)
    b = Engine(20, 48, 3, True, 'concat', 9, learning_rate=1e-2)
    a.set_params(p0)
    b.set_params(p0)
    assert a.comm_size() == 1
    uid = a.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    a.comm_init(uid, 0, 1)
    with pytest.raises(_lib.NasrError, match='already has a communicator'):
        a.comm_init(uid, 0, 1)
    assert len(a.grad_buckets()) == 3
    for _ in range(3):
        a.upload_batch(feats, seq_len, labels, label_len)
        a.compute_grads()
        a.comm_allreduce_grads()
        a.apply_adam(1.0 / a.comm_size())
        la = a.get_loss()
        lb = b.train_step(feats, seq_len, labels, label_len)
        assert la == lb
    np.testing.assert_array_equal(a.get_params(), b.get_params())
    assert a.comm_mean([la, 0.25]) == pytest.approx([la, 0.25])
    a.comm_destroy()
    assert a.comm_size() == 1
    a.close()
    b.close()


@needs_persistent
def test_begin_finish_and_async_train_follow_the_synchronous_path(tmp_path):
    """train() = begin_step() + finish_step(); with async_step the values come back after the forward pass + CTC while the
    backward pass still runs.  Losses, LERs and parameters are those of the synchronous path (async_step = False), bit for
    bit, over several steps with a validation in between."""
    nets = []
    for k, fast in enumerate((True, False)):
        cfg = Config(make_config(tmp_path, network='networks.bilstm_ctc_net.BiLstm3x500CTCNet', num_gpus='1', batch_size='4',
                                 model_dir=str(tmp_path / ('a%d' % k))), True)
        net = cfg.load_network(fortraining=True)
        net.async_step = fast
        nets.append((net, DataSet(cfg.train_input, cfg).get_next_batch()))
    (a, batch), (b, _) = nets
    b.engine.set_params(a.engine.get_params())
    outs = []
    for step in range(4):
        if step == 2:
            a.begin_step(*batch)
            ra = a.finish_step()
        else:
            ra = a.train(*batch)
        rb = b.train(*batch)
        assert ra == rb, (step, ra, rb)
        if step == 2:
            assert a.validate(*batch) == b.validate(*batch)       # settles the pending step first
        outs.append(ra)
    assert a.global_step == b.global_step == 4
    np.testing.assert_array_equal(a.engine.get_params(), b.engine.get_params())
    assert a.engine.get_adam_state()[2] == b.engine.get_adam_state()[2] == 4


@needs_persistent
def test_async_train_repeats_a_step_whose_backward_pass_was_void(tmp_path, monkeypatch, caplog):
    """A persistent-BPTT abort (injected into the backward kernel only) is not known when train() returns - the loss of the
    forward pass is good, the step is void.  The next call notices it (the previous step's fault word, read without a
    stream sync) and repeats that batch: after two calls the parameters have had two updates, like an undisturbed run."""
    import logging
    cfgs = [Config(make_config(tmp_path, num_gpus='1', model_dir=str(tmp_path / ('v%d' % k))), True) for k in range(2)]
    net, ref = [c.load_network(fortraining=True) for c in cfgs]
    ref.engine.set_params(net.engine.get_params())
    batch = DataSet(cfgs[0].train_input, cfgs[0]).get_next_batch()
    want = [ref.train(*batch) for _ in range(2)]
    monkeypatch.setenv('NASR_PERSIST_FAULT', '2')
    monkeypatch.setenv('NASR_PERSIST_FAULT_KERNEL', 'bwd')
    got0 = net.train(*batch)
    monkeypatch.delenv('NASR_PERSIST_FAULT')
    monkeypatch.delenv('NASR_PERSIST_FAULT_KERNEL')
    assert float(got0[0]) == pytest.approx(float(want[0][0]), rel=2e-6)       # the forward pass of the void step was fine
    with caplog.at_level(logging.WARNING):
        got1 = net.train(*batch)                  # step 2 on the still un-updated parameters, then the repeat of step 1
    assert any('void' in r.getMessage() for r in caplog.records)
    assert float(got1[0]) == pytest.approx(float(want[0][0]), rel=2e-5)       # its loss is that of un-updated parameters
    net.save_checkpoint()                         # settles the last step
    assert net.engine.recurrence_mode == 'per-step' and net.global_step == 2
    assert net.engine.get_adam_state()[2] == 2
    np.testing.assert_allclose(net.engine.get_params(), ref.engine.get_params(), rtol=0, atol=1e-4)


@needs_persistent
def test_async_train_repeats_both_steps_when_the_one_in_flight_was_void_too(tmp_path, monkeypatch, caplog):
    """What voids step N (a co-tenant, a placement) is still there when step N+1 is enqueued - the host learns about N
    one call later.  With the BPTT fault injected across two train() calls both steps are void: the second call must
    learn the fate of the step in flight (nasr_settle_token) before it enqueues anything, then repeat N and N+1 in order -
    three calls leave three updates, like an undisturbed run, and the second call reports the repeat's values."""
    import logging
    cfgs = [Config(make_config(tmp_path, num_gpus='1', model_dir=str(tmp_path / ('w%d' % k))), True) for k in range(2)]
    net, ref = [c.load_network(fortraining=True) for c in cfgs]
    ref.engine.set_params(net.engine.get_params())
    ds = DataSet(cfgs[0].train_input, cfgs[0])
    b0 = ds.get_next_batch()
    b1 = ds.get_next_batch() if ds.has_more_batches() else b0
    want = [ref.train(*b) for b in (b0, b1, b0)]
    monkeypatch.setenv('NASR_PERSIST_FAULT', '2')
    monkeypatch.setenv('NASR_PERSIST_FAULT_KERNEL', 'bwd')
    got0 = net.train(*b0)                         # step 1: BPTT aborts, nobody knows yet
    with caplog.at_level(logging.WARNING):
        got1 = net.train(*b1)                     # step 2 runs into the same fault; then: settle 1, settle 2, redo 1, redo 2
    monkeypatch.delenv('NASR_PERSIST_FAULT')
    monkeypatch.delenv('NASR_PERSIST_FAULT_KERNEL')
    msgs = [r.getMessage() for r in caplog.records]
    assert any('step 1 was void' in m for m in msgs) and any('step 2 was void as well' in m for m in msgs)
    assert net.engine.get_adam_state()[2] == 2 and net.engine.recurrence_mode == 'per-step'
    assert float(got0[0]) == pytest.approx(float(want[0][0]), rel=2e-6)
    assert float(got1[0]) == pytest.approx(float(want[1][0]), rel=2e-5)       # the repeat's loss: after step 1's update
    got2 = net.train(*b0)
    assert float(got2[0]) == pytest.approx(float(want[2][0]), rel=2e-4)
    net.save_checkpoint()
    assert net.global_step == 3 and net.engine.get_adam_state()[2] == 3
    np.testing.assert_allclose(net.engine.get_params(), ref.engine.get_params(), rtol=0, atol=1e-4)


def test_step_tokens_name_the_optimiser_steps(tmp_path):
    """include/nasr.h: nasr_step_token / nasr_settle_token - a token per nasr_apply_adam, the last four remembered."""
    from neuralasr_amd import _lib
    cfg = Config(make_config(tmp_path, num_gpus='1'), True)
    net = cfg.load_network(fortraining=True)
    batch = DataSet(cfg.train_input, cfg).get_next_batch()
    e = net.engine
    assert e.step_token() == 0
    toks = []
    for _ in range(6):
        net.begin_step(*batch)
        toks.append(net._begun[1][1])
        net.finish_step()
    assert toks == [1, 2, 3, 4, 5, 6] and e.step_token() == 6
    assert [e.settle_token(t) for t in toks[2:]] == [False] * 4
    with pytest.raises(_lib.NasrError, match='older'):
        e.settle_token(2)
    with pytest.raises(_lib.NasrError, match='no such step'):
        e.settle_token(7)


def test_train_ler_is_the_beam_search_of_the_steps_own_logits_lazily_or_not(tmp_path):
    """tfnetwork.py:188-189 fetches mean_ler = the LER of the width-100 beam decode with every step.  HipNetwork decodes the
    step's own logits (copied out behind the CTC kernels) on host threads: train() waits for it, finish_step(lazy=True)
    hands back a handle that train_model resolves when it logs - same numbers either way, equal to the oracle's beam LER of
    the parameters the step started from; 'greedy' switches to the device's greedy decode."""
    cfg = Config(make_config(tmp_path, num_gpus='1', batch_size='4'), True)
    net = cfg.load_network(fortraining=True)
    assert net.train_ler_decoder == 'beam'
    spec = spec_of(net, cfg)
    ds = DataSet(cfg.train_input, cfg)
    batch = ds.get_next_batch()
    mfccs, labels, seq_len, labels_len = batch
    lens = [int(s) for s in seq_len]

    def oracle_lers():
        params = [p.astype(np.float64) for p in O.unflatten(spec, net.engine.get_params())]
        lg, _ = O.network_forward(spec, params, mfccs, lens)
        beam = [O.ctc_beam_search(lg[:lens[b], b], 100, True)[0] for b in range(len(lens))]
        return (O.label_error_rate(beam, labels, labels_len), O.label_error_rate(O.greedy_decode(lg, lens), labels, labels_len))
    want_beam, _ = oracle_lers()
    _, ler = net.train(*batch)
    assert float(ler) == pytest.approx(want_beam, abs=1e-6)
    net.save_checkpoint()                                   # settles the step: the parameters are those after one update
    want_beam, _ = oracle_lers()
    net.begin_step(*batch)
    loss, handle = net.finish_step(lazy=True)
    assert hasattr(handle, 'result') and isinstance(loss, np.float32)
    assert float(handle.result()) == pytest.approx(want_beam, abs=1e-6)
    net.save_checkpoint()
    net.train_ler_decoder = 'greedy'
    _, want_greedy = oracle_lers()
    _, ler = net.train(*batch)
    assert float(ler) == pytest.approx(want_greedy, abs=1e-6)


def test_train_model_logs_the_beam_ler_of_every_step(tmp_path, caplog):
    """The log line of train.py:32-34 averages the steps' mean_ler: with the lazily decoded LERs it must show the same number
    as a loop that waits for every step's beam search."""
    import logging
    from neuralasr_amd import train as train_mod
    cfgs = [Config(make_config(tmp_path, num_gpus='1', batch_size='2', report_step='3', epochs='2',
                               model_dir=str(tmp_path / ('lz%d' % k))), True) for k in range(2)]
    ref = cfgs[1].load_network(fortraining=True)
    ds = DataSet(cfgs[1].train_input, cfgs[1])
    lers = []
    for _ in range(cfgs[1].epochs):
        while ds.has_more_batches():
            lers.append(float(ref.train(*ds.get_next_batch())[1]))
        ds.reset_epoch()
    with caplog.at_level(logging.INFO, logger='NeuralASR'):
        train_mod.train_model(DataSet(cfgs[0].train_input, cfgs[0]), None, cfgs[0])
    lines = [r.getMessage() for r in caplog.records if r.getMessage().startswith('Step: ')]
    assert len(lines) == len(lers) // 3 and len(lines) >= 1
    for i, ln in enumerate(lines):
        assert ', ler = %.4f' % np.mean(lers[3 * i:3 * i + 3]) in ln, (ln, lers)


def test_step_results_with_logits_only():
    """nasr_set_step_decode(h, 2): the step's loss, fault word and logits behind the CTC kernels, no greedy decode (what the
    beam-LER train step uses) - same loss and logits as mode 3, empty hypotheses, and the gradients are not disturbed."""
    from neuralasr_amd.engine import Engine
    spec = O.ModelSpec(9, 24, 2, True, 'concat', 7)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 5, 23, seed=3, var_len=True, Lmin=1, Lmax=5)
    e = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes)
    e.set_params(O.flatten(O.init_params(spec, seed=2)))
    e.upload_batch(feats, seq_len, labels, label_len)
    out = {}
    for greedy in (True, False):
        e.set_step_decode(True, logits=True, greedy=greedy)
        e.compute_grads()
        loss, fault, hyps = e.step_results(5, 23)
        out[greedy] = (loss, fault, hyps, e.step_logits(5, 23), e.get_grads())
    assert out[True][0] == out[False][0] and out[True][1] == out[False][1] == 0
    np.testing.assert_array_equal(out[True][3], out[False][3])
    np.testing.assert_array_equal(out[True][4], out[False][4])
    np.testing.assert_allclose(out[True][3], e.forward(feats, seq_len), atol=1e-6)
    assert any(len(h) for h in out[True][2]) and not any(len(h) for h in out[False][2])
    assert out[True][2] == e.greedy_decode(feats, seq_len)
    e.close()
