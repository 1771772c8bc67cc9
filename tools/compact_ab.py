"""A/B of the ragged-batch row compaction (nasr_ctx.h: cmp_rows): the bench step (3x500 BiLSTM, B=16, T=500) on batches whose
lengths are U{f T .. T} (one utterance at T), with NASR_COMPACT=0 and =1, each in its own process.
   python tools/compact_ab.py            # the table
"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def child(frac, B, T):
    import torch
    import bench
    from neuralasr_amd.engine import Engine
    spec, _ = bench.workload_spec('bilstm3x500')
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    e = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
               learning_rate=1e-4, device_id=0, stream=st.cuda_stream)
    e.set_graph_mode(True)
    e.set_params(bench.init_params(e.tensors(), seed=1))
    feats, seq_len, labels, label_len = bench.synth_batch(spec, B, T, seed=1234, var_len=True)
    rs = np.random.RandomState(5)
    seq_len = np.sort(rs.randint(max(int(T * frac), 1), T + 1, size=B)).astype(np.int32)
    seq_len[-1] = T
    for b in range(B):
        feats[b, seq_len[b]:] = 0
    label_len = np.minimum(label_len, np.maximum(seq_len // 2, 1)).astype(np.int32)
    e.upload_batch(feats, seq_len, labels, label_len)
    for _ in range(10):
        e.compute_grads(); e.apply_adam(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        e.compute_grads(); e.apply_adam(1.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f'RESULT {int(seq_len.sum())} {ms:.4f} {e.get_loss():.6f}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    print('B T frac frames(valid/all) ms_off ms_on loss_off loss_on')
    for B, T in ((16, 500), (32, 500)):
        for frac in (0.75, 0.5, 0.25, 0.05):
            res = []
            for on in ('0', '1'):
                env = dict(os.environ, NASR_COMPACT=on)
                out = subprocess.run([sys.executable, __file__, 'child', str(frac), str(B), str(T)], env=env, capture_output=True,
                                     text=True, timeout=300)
                line = [l for l in out.stdout.splitlines() if l.startswith('RESULT')]
                if not line:
                    print(out.stdout[-2000:], out.stderr[-2000:])
                    sys.exit(1)
                res.append(line[0].split()[1:])
            print(B, T, frac, f'{res[0][0]}/{B * T}', res[0][1], res[1][1], res[0][2], res[1][2], flush=True)
