"""GPU, two processes: the N > 1 protocol of include/nasr.h ("data-parallel building blocks") end to end on real
engines - two ranks (both on the one GPU of the test box, `gloo` carrying the device tensors: RCCL refuses two ranks on
one device) each upload half of the global batch, compute gradients, sum the gradient buffer BUCKET BY BUCKET in the
order nasr_grad_bucket gives (after nasr_grad_bucket_wait on a side stream), and apply Adam with 1/n.  Both ranks must
end with identical parameters, equal to one engine stepping on the global batch (the concat merge is self-consistent
under sharding), and the fault word at the head of the buffer must stay zero.

The per-step kernels are forced (NASR_PERSIST=0): a persistent launch wants every CU of the device, and two processes
launching them on ONE GPU at the same time is exactly the sharing the census cannot promise (include/nasr.h)."""
import os
import socket

import numpy as np
import pytest

from oracle import nasr_oracle as O

pytestmark = pytest.mark.gpu

SPEC = O.ModelSpec(18, 40, 2, True, 'concat', 9)
B, T, STEPS = 8, 26, 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch():
    return O.synth_batch(SPEC, B, T, seed=41, var_len=True, Lmin=1, Lmax=5)


def _start_params():
    rs = np.random.RandomState(3)
    return O.flatten([p + 0.05 * rs.randn(*p.shape) for p in O.init_params(SPEC, seed=3)]).astype(np.float32)


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['NASR_PERSIST'] = '0'
    import torch
    import torch.distributed as dist
    from neuralasr_amd.engine import Engine
    from neuralasr_amd.parallel import take_shard
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ts = torch.cuda.Stream()
        torch.cuda.set_stream(ts)
        side = torch.cuda.Stream()
        e = Engine(SPEC.feature_size, SPEC.hidden, SPEC.num_layers, True, 'concat', SPEC.num_classes, learning_rate=1e-3,
                   stream=ts.cuda_stream)
        e.set_params(_start_params())
        feats, seq_len, labels, label_len = _batch()
        f, l, s, ll = take_shard(feats, labels, list(seq_len), list(label_len), world, rank)
        gt = e.grad_tensor()
        buckets = e.grad_buckets()
        assert len(buckets) == SPEC.num_layers and sorted(buckets)[0][0] == 0
        assert sum(c for _, c in buckets) == gt.numel()
        losses = []
        for _ in range(STEPS):
            e.upload_batch(f, s, l, ll)
            e.compute_grads()
            for i, (o, c) in enumerate(buckets):           # the exchange, bucket by bucket, in completion order
                with torch.cuda.stream(side):
                    e.bucket_wait(i, side.cuda_stream)
                side.synchronize()                         # gloo stages device tensors through the host
                dist.all_reduce(gt[o:o + c], op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            e.apply_adam(1.0 / world)
            assert not e.step_void()
            losses.append(e.get_loss())
        np.savez(os.path.join(out_dir, 'r%d.npz' % rank), params=e.get_params(), losses=np.array(losses),
                 head=gt[:32].cpu().numpy())
        e.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_global_batch(tmp_path):
    import torch.multiprocessing as mp
    from neuralasr_amd.engine import Engine
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'r0.npz'), np.load(tmp_path / 'r1.npz')
    np.testing.assert_array_equal(r0['params'], r1['params'])        # replicas never diverge
    assert not r0['head'].any() and not r1['head'].any()             # fault word and its padding stay zero
    # one engine on the global batch
    feats, seq_len, labels, label_len = _batch()
    os.environ['NASR_PERSIST'] = '0'
    try:
        ref = Engine(SPEC.feature_size, SPEC.hidden, SPEC.num_layers, True, 'concat', SPEC.num_classes, learning_rate=1e-3)
    finally:
        del os.environ['NASR_PERSIST']
    ref.set_params(_start_params())
    ref_losses = [ref.train_step(feats, seq_len, labels, label_len) for _ in range(STEPS)]
    # reported loss = mean of the shard means = the global mean for equal shards
    np.testing.assert_allclose((r0['losses'] + r1['losses']) / 2, ref_losses, rtol=2e-6)
    # Adam divides by sqrt(v): fp32 summation-order differences of near-zero gradients are amplified up to the step size
    np.testing.assert_allclose(r0['params'], ref.get_params(), rtol=0, atol=5e-5)
    assert (np.abs(r0['params'] - ref.get_params()) > 2e-6).mean() < 2e-3
    ref.close()


def _net_worker(rank, world, port, cfg_path, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), NASR_PERSIST='0', LOCAL_RANK='0', RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from neuralasr_amd.config import Config
    from neuralasr_amd.dataset import DataSet
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        cfg = Config(cfg_path, True)
        cfg.model_dir = os.path.join(out_dir, 'model')              # rank 0 wipes / writes it
        net = cfg.load_network(fortraining=True)
        assert net.coll.world == 2 and net._towers() == (2, [rank])
        batch = DataSet(cfg.train_input, cfg).get_next_batch()      # every rank sees the GLOBAL batch (train.py)
        outs = [net.train(*batch) for _ in range(3)]
        v = net.validate(*batch)
        np.savez(os.path.join(out_dir, 'n%d.npz' % rank), params=net.engine.get_params(), outs=np.array(outs, np.float64),
                 valid=np.array(v, np.float64), step=net.engine.get_adam_state()[2])
    finally:
        dist.destroy_process_group()


def test_hipnetwork_train_with_one_process_per_tower(tmp_path):
    """HipNetwork.train / validate under torch.distributed with one process per tower (two ranks on the one GPU, gloo):
    the asynchronous step (results after the forward pass, gradient all-reduce + Adam(1/n) enqueued, loss / LER averaged
    over the ranks) gives the parameters and the values of ONE process time-slicing the same two towers."""
    import torch.multiprocessing as mp
    from test_gpu_network import make_config
    from neuralasr_amd.config import Config
    from neuralasr_amd.dataset import DataSet
    cfg_path = make_config(tmp_path, network='networks.lstm_ctc_net.SmallLstmCTCNet')
    mp.spawn(_net_worker, args=(2, _free_port(), cfg_path, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'n0.npz'), np.load(tmp_path / 'n1.npz')
    np.testing.assert_array_equal(r0['params'], r1['params'])
    np.testing.assert_array_equal(r0['outs'], r1['outs'])
    assert int(r0['step']) == int(r1['step']) == 3
    os.environ['NASR_PERSIST'] = '0'
    try:
        cfg = Config(cfg_path, True)
        cfg.model_dir = str(tmp_path / 'model_ref')
        ref = cfg.load_network(fortraining=True)                    # num_gpus = 2: both towers in this process
    finally:
        del os.environ['NASR_PERSIST']
    batch = DataSet(cfg.train_input, cfg).get_next_batch()
    want = np.array([ref.train(*batch) for _ in range(3)], np.float64)
    np.testing.assert_allclose(r0['outs'], want, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(r0['valid'], np.array(ref.validate(*batch), np.float64), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(r0['params'], ref.engine.get_params(), rtol=0, atol=2e-4)
