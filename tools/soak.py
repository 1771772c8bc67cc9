"""Soak of the training step: N optimiser steps of the bench workload on one resident batch, twice from the same start;
the two runs must end in bitwise the same parameters, with no persistent-recurrence abort on the way.
   python tools/soak.py [steps=1500] [var_len=0]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench                                   # noqa: E402
from neuralasr_amd.engine import Engine        # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    var_len = len(sys.argv) > 2 and sys.argv[2] == '1'
    spec, name = bench.workload_spec('bilstm3x500')
    feats, seq_len, labels, label_len = bench.synth_batch(spec, 16, 500, seed=1234, var_len=var_len)
    ends = []
    for run in range(2):
        e = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
                   learning_rate=1e-4)
        e.set_graph_mode(True)
        e.set_params(bench.init_params(e.tensors(), seed=1))
        e.upload_batch(feats, seq_len, labels, label_len)
        t0 = time.perf_counter()
        losses = []
        for i in range(steps):
            e.compute_grads()
            e.apply_adam(1.0)
            if (i + 1) % 250 == 0:
                losses.append(e.get_loss())
        e.synchronize()
        dt = time.perf_counter() - t0
        p = e.get_params()
        ends.append(p)
        print(f'run {run}: {name}, var_len={var_len}, {steps} steps in {dt:.2f} s ({dt / steps * 1e3:.3f} ms/step), rows {e.resident_rows()}, '
              f'loss every 250 steps {" ".join("%.4f" % l for l in losses)}, persist aborts/rearms {e.persist_stats()}, '
              f'mode {e.recurrence_mode}, finite {bool(np.isfinite(p).all())}', flush=True)
        e.close()
    print('bitwise equal end parameters:', bool(np.array_equal(ends[0], ends[1])))


if __name__ == '__main__':
    main()
