"""End-to-end train() cost from host buffers (upload every step), vs the resident-batch step bench.py times."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import nasr_oracle as O
from neuralasr_amd.engine import Engine
for name, spec in (('literal', O.ModelSpec(546, 500, 1, True, 'stack_reshape', 29)), ('3x500', O.ModelSpec(546, 500, 3, True, 'concat', 29))):
    feats, seq_len, labels, label_len = O.synth_batch(spec, 16, 500, seed=1)
    e = Engine(546, 500, spec.num_layers, True, spec.merge, 29)
    e.set_params(O.flatten(O.init_params(spec, 1)))
    for _ in range(3): e.train_step(feats, seq_len, labels, label_len)
    t0 = time.perf_counter()
    for _ in range(10): e.train_step(feats, seq_len, labels, label_len)
    t1 = time.perf_counter()
    e.upload_batch(feats, seq_len, labels, label_len)
    e.synchronize()
    t2 = time.perf_counter()
    for _ in range(10): e.upload_batch(feats, seq_len, labels, label_len)
    e.synchronize()
    t3 = time.perf_counter()
    e.set_step_decode(True)
    for _ in range(10):
        e.train_step(feats, seq_len, labels, label_len); hy = e.get_decoded(16, 500); e.label_error_rate(hy, labels, label_len)
    t4 = time.perf_counter()
    print(name, 'train_step from host: %.2f ms; upload alone: %.2f ms; train_step + greedy LER: %.2f ms' % ((t1 - t0) * 100, (t3 - t2) * 100, (t4 - t3) * 100))
    e.close()
