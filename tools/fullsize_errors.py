#!/usr/bin/env python3
"""Loss and gradient errors of the literal reference net at BASELINE.json's full size (B 16, T 500, F 546, H 500, C 29)
against the fp64 oracle, with the forward recurrence on fp16 planes (default) and on fp32 MFMAs (NASR_REC=f32).
    python tools/fullsize_errors.py            (runs itself once per mode)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)


def one(mode):
    from neuralasr_amd.engine import Engine
    from oracle import nasr_oracle as O
    spec = O.ModelSpec(546, 500, 1, True, 'stack_reshape', 29)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 16, 500, seed=1234, var_len=True)
    params = O.init_params(spec, seed=1)
    e = Engine(546, 500, 1, True, 'stack_reshape', 29)
    e.set_params(O.flatten(params))
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    ref = '/tmp/fullsize_oracle.npz'
    if not os.path.exists(ref):
        p32 = [p.astype(np.float32).astype(np.float64) for p in params]
        loss_o, nll_o, grads_o, _ = O.network_loss_and_grads(spec, p32, feats, seq_len, labels, label_len)
        np.savez(ref, loss=loss_o, nll=nll_o, g=O.flatten(grads_o))
    z = np.load(ref)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    per = {n: rel(grads[o:o + r * c], z['g'][o:o + r * c]) for n, o, r, c in e.tensors()}
    print(f"{mode:5s} loss rel err {abs(loss - float(z['loss'])) / float(z['loss']):.1e}  worst nll rel err "
          f"{float(np.max(np.abs(nll - z['nll']) / z['nll'])):.1e}  gradient rel-L2 {rel(grads, z['g']):.1e}  per tensor "
          + ' '.join(f'{k}={v:.1e}' for k, v in per.items()))


if __name__ == '__main__':
    if len(sys.argv) > 1:
        one(sys.argv[1])
    else:
        for mode in ('f16', 'f32'):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), mode], env=dict(os.environ, NASR_REC=mode))
