#!/usr/bin/env python3
"""One GPU: does work on a side stream, released bucket by bucket (nasr_grad_bucket_wait), co-run with the persistent
BPTT of the layers below without tripping its bounded spins, and what does it cost?  The side work stands in for the
RCCL kernels of the bucketed all-reduce (which a single GPU cannot run): an in-place elementwise pass over the bucket
(far MORE workgroups than an RCCL kernel uses - an upper bound on the interference).
    python tools/overlap_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from neuralasr_amd.engine import Engine          # noqa: E402
from oracle import nasr_oracle as O              # noqa: E402

spec = O.ModelSpec(546, 500, 3, True, 'concat', 29)
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
eng = Engine(546, 500, 3, True, 'concat', 29, learning_rate=1e-4, stream=ts.cuda_stream)
eng.set_params(O.flatten(O.init_params(spec, seed=1)).astype(np.float32))
feats, seq_len, labels, label_len = O.synth_batch(spec, 16, 500, seed=1)
eng.upload_batch(feats, seq_len, labels, label_len)
gt = eng.grad_tensor()
views = [gt[o:o + c] for o, c in eng.grad_buckets()]
side = torch.cuda.Stream()


def step(mode):
    eng.compute_grads()
    if mode == 'side':
        with torch.cuda.stream(side):
            for i, v in enumerate(views):
                eng.bucket_wait(i, side.cuda_stream)
                v.mul_(1.0)
        ts.wait_stream(side)
    elif mode == 'tail':
        for v in views:
            v.mul_(1.0)
    eng.apply_adam(1.0)


for mode in ('none', 'tail', 'side', 'none', 'tail', 'side'):
    for _ in range(3):
        step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step(mode)
    torch.cuda.synchronize()
    print(f'{mode:5s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step  recurrence {eng.recurrence_mode}  void {eng.step_void()}')
