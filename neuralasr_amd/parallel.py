"""Data-parallel sharding: the reference's in-graph tower replication (make_parallel + average_gradients,
networks/tfnetwork.py:72-140) re-expressed as ONE SHARD PER GPU / PROCESS with a sum all-reduce of one flat
fp32 gradient buffer and the 1/n folded into the optimiser step.

  * the global batch (config.batch_size = per-GPU batch x num_gpus, config.py:35-36) is cut into n equal
    contiguous chunks on axis 0 (tf.split semantics); every shard keeps the GLOBAL max T, so the literal
    BiLstmCTCNet's stack-reshape index map (a function of the shard's B and of T) is what the
    reference's towers see;
  * loss / LER reported = mean of the shard means (tfnetwork.py:135-136);
  * gradient = mean over shards of each shard's gradient of its shard-mean loss (tfnetwork.py:72-86).

The collective is torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests); torch is plumbing only."""
import numpy as np


def shard_bounds(global_batch, world, rank):
    """Rows [lo, hi) of the global batch that tower `rank` of `world` owns (tf.split, axis 0)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError('bad world/rank %r/%r' % (world, rank))
    if global_batch % world:
        raise ValueError('global batch %d does not split evenly over %d towers' % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def take_shard(mfccs, labels, seq_len, labels_len, world, rank):
    lo, hi = shard_bounds(len(seq_len), world, rank)
    labels = np.asarray(labels)
    return (np.asarray(mfccs)[lo:hi], labels[lo:hi] if labels.ndim else labels, list(seq_len[lo:hi]),
            list(labels_len[lo:hi]) if labels_len is not None else None)


class Collective:
    """Thin view of the process group: world size, rank, sum-all-reduce of tensors / python floats."""

    def __init__(self):
        try:
            import torch.distributed as dist
            self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        except Exception:
            self.dist = None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        # loss / LER means travel as host floats.  Under an NCCL process group they get a gloo group of their own: an
        # NCCL collective would queue up behind the gradient buckets of the step (one communicator, issue order), i.e.
        # wait for the END of the backward pass, and NCCL takes no CPU tensors.  (Collective: every rank constructs this.)
        self.scalar_group = None
        if self.dist and self.world > 1 and self.dist.get_backend() == 'nccl':
            self.scalar_group = self.dist.new_group(backend='gloo')

    def all_reduce_sum_(self, tensor):
        if self.dist:
            self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM)
        return tensor

    def bucketed(self, engine, grad_tensor):
        """A BucketedAllReduce over this process group (None when the buffer is a single bucket or the backend
        cannot run collectives on device streams)."""
        if not self.dist or self.dist.get_backend() != 'nccl' or len(engine.grad_buckets()) < 2:
            return None
        return BucketedAllReduce(engine, self.dist, grad_tensor)

    def mean_scalars(self, values, device=None):
        """mean over ranks of a few python floats (loss, ler): the reduce_mean of tfnetwork.py:135-136."""
        if not self.dist:
            return [float(v) for v in values]
        import torch
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device or 'cpu')
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.scalar_group if device is None else None)
        return [float(x) / self.world for x in t.tolist()]


class BucketedAllReduce:
    """Sum all-reduce of an engine's gradient buffer, bucket by bucket, on a side stream that waits for each bucket's
    completion event: the exchange of the upper layers' gradients runs under the BPTT and weight-gradient GEMMs of the
    layers below (the reference's average_gradients, tfnetwork.py:72-86, can only start when every tower's backward pass
    has ended).  Usage per step, with the engine's stream current:

        engine.compute_grads(); reducer.all_reduce(); engine.apply_adam(1 / world)

    all_reduce() returns once the collectives are ENQUEUED; the current stream has been made to wait for them.
    xGMI is point-to-point and a ring is per-link bound, so the buckets stay large: one per LSTM layer (8-21 MB at
    2x500), never per tensor."""

    def __init__(self, engine, dist, grad_tensor=None):
        import torch
        self.torch, self.dist, self.engine = torch, dist, engine
        self.tensor = engine.grad_tensor() if grad_tensor is None else grad_tensor
        self.buckets = engine.grad_buckets()
        covered = sorted(self.buckets)
        if covered[0][0] != 0 or sum(c for _, c in covered) != self.tensor.numel() or \
                any(a[0] + a[1] != b[0] for a, b in zip(covered, covered[1:])):
            raise RuntimeError('gradient buckets do not tile the gradient buffer: %r' % (self.buckets,))
        self.views = [self.tensor[o:o + c] for o, c in self.buckets]
        self.stream = torch.cuda.Stream(device=self.tensor.device)

    def all_reduce(self):
        works = []
        with self.torch.cuda.stream(self.stream):
            for i, v in enumerate(self.views):
                self.engine.bucket_wait(i, self.stream.cuda_stream)
                works.append(self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()            # orders the CURRENT stream (the engine's) behind the collective; no host sync
        return self.tensor

