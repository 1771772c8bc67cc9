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
