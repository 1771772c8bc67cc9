"""The N > 1 path on CPU: world_size-2 `gloo` ranks run the host-side sharding (neuralasr_amd.parallel)
with the oracle standing in for the per-GPU engine, and must reproduce the reference's tower arithmetic
(make_parallel + average_gradients, networks/tfnetwork.py:72-140)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neuralasr_amd.parallel import Collective, shard_bounds, take_shard
from oracle import nasr_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, spec, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        coll = Collective()
        assert coll.world == world and coll.rank == rank
        feats, seq_len, labels, label_len = O.synth_batch(spec, 4, 9, seed=11, var_len=True, Lmin=1, Lmax=3)
        params = O.init_params(spec, seed=4)
        f, l, s, ll = take_shard(feats, labels, [np.int32(x) for x in seq_len], list(label_len), world, rank)
        loss, _, grads, _ = O.network_loss_and_grads(spec, params, f, s, l, ll)     # this rank's tower
        g = torch.tensor(O.flatten(grads))
        coll.all_reduce_sum_(g)                                                     # the RCCL step, on gloo
        g = g.numpy() / world                                                       # 1/n folded into Adam
        mean_loss, = coll.mean_scalars([loss])
        np.savez(os.path.join(out_dir, 'r%d.npz' % rank), g=g, loss=mean_loss)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('merge', ['stack_reshape', 'concat'])
def test_two_gloo_ranks_reproduce_tower_averaging(tmp_path, merge):
    spec = O.ModelSpec(4, 3, 1, True, merge, 5)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, spec, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'r0.npz'), np.load(tmp_path / 'r1.npz')
    np.testing.assert_array_equal(r0['g'], r1['g'])            # every rank holds the same averaged gradient
    feats, seq_len, labels, label_len = O.synth_batch(spec, 4, 9, seed=11, var_len=True, Lmin=1, Lmax=3)
    params = O.init_params(spec, seed=4)
    loss_ref, grads_ref = O.data_parallel_loss_and_grads(spec, params, feats, seq_len, labels, label_len, 2)
    np.testing.assert_allclose(r0['g'], O.flatten(grads_ref), atol=1e-14)
    assert float(r0['loss']) == pytest.approx(loss_ref, rel=1e-13)
    if merge == 'concat':       # self-consistent merge: n towers == one tower on the global batch
        l1, _, g1, _ = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
        np.testing.assert_allclose(r0['g'], O.flatten(g1), atol=1e-13)
        assert float(r0['loss']) == pytest.approx(l1, rel=1e-13)


def test_shard_bounds_are_tf_split():
    assert [shard_bounds(16, 4, r) for r in range(4)] == [(0, 4), (4, 8), (8, 12), (12, 16)]
    assert shard_bounds(3, 1, 0) == (0, 3)
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0)          # tf.split needs an even split
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def test_take_shard_keeps_global_T_and_types():
    feats = np.zeros((4, 7, 3), np.float32)
    labels = np.arange(8).reshape(4, 2)
    f, l, s, ll = take_shard(feats, labels, [np.int32(7), np.int32(5), np.int32(6), np.int32(2)], [2, 1, 2, 1], 2, 1)
    assert f.shape == (2, 7, 3) and l.tolist() == [[4, 5], [6, 7]] and [int(x) for x in s] == [6, 2] and ll == [2, 1]


def test_single_process_collective_is_identity():
    c = Collective()
    assert c.world == 1 and c.rank == 0
    assert c.mean_scalars([1.5, 2.0]) == [1.5, 2.0]
