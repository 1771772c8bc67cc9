#!/usr/bin/env python3
"""One GPU: what a CU-resident collective costs the step it co-runs with.  RCCL cannot run two ranks on one device, so a
kernel SHAPED like a ring all-reduce step stands in for it (nasr_diag_bucket_traffic: nblocks workgroups x 256 threads, each
holding its CU while it sweeps its slice of a gradient bucket `passes` times with 16-byte loads and stores), released on a
side stream by the bucket events exactly as parallel.BucketedAllReduce releases the real collectives.  Reported per setting:
ms per step, the persistent-recurrence abort count and the recurrence mode afterwards, with the bucket events held back over
the next persistent BPTT launch (defer, the default) and released at once (eager).
    python tools/rccl_standin.py [--blocks 32 64] [--passes 4 12] [--steps 30]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench                                     # noqa: E402  (workload + synthetic batch of the default bench line)
from neuralasr_amd.engine import Engine          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--blocks', type=int, nargs='+', default=[32, 64])
ap.add_argument('--passes', type=int, nargs='+', default=[4, 12])
ap.add_argument('--steps', type=int, default=30)
args = ap.parse_args()

spec, wname = bench.workload_spec('bilstm3x500')
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
eng = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
             learning_rate=1e-4, stream=ts.cuda_stream)
eng.set_params(bench.init_params(eng.tensors(), seed=1))
feats, seq_len, labels, label_len = bench.synth_batch(spec, 16, 500, seed=1234)
eng.upload_batch(feats, seq_len, labels, label_len)
nb = len(eng.grad_buckets())
side = torch.cuda.Stream()
print(f'{wname}: {nb} gradient buckets of {[c * 4 // 2**20 for _, c in eng.grad_buckets()]} MiB')


def step(blocks, passes):
    eng.compute_grads()
    if blocks:
        for i in range(nb):
            eng.diag_bucket_traffic(i, side.cuda_stream, blocks, passes)
        ts.wait_stream(side)
    eng.apply_adam(1.0)


def run(defer, blocks, passes):
    eng.set_bucket_defer(defer)
    a0, _ = eng.persist_stats()
    for _ in range(3):
        step(blocks, passes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(blocks, passes)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    # the stand-in alone, to know how long it holds its CUs
    alone = None
    if blocks:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            for i in range(nb):
                eng.diag_bucket_traffic(i, side.cuda_stream, blocks, passes)
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t1) / 5 * 1e3
    void = eng.step_void()
    a1, _ = eng.persist_stats()
    print(f"{'defer' if defer else 'eager'}  blocks {blocks:3d}  passes {passes:3d}:  {ms:7.3f} ms/step"
          + (f'  (stand-in alone {alone:6.3f} ms for all buckets)' if alone else '')
          + f'  aborts {a1 - a0}  void {void}  recurrence {eng.recurrence_mode}', flush=True)
    return ms


for rep in range(2):
    base = run(True, 0, 0)
    for defer in (True, False):
        for blocks in args.blocks:
            for passes in args.passes:
                run(defer, blocks, passes)
eng.close()
