"""Times nasr_ctc_beam_search (tf.nn.ctc_beam_search_decoder defaults, networks/tfnetwork.py:61-64) on the GPU box host cores at
B 16, width 100: random-init posteriors and posteriors after 60 training steps, 3x500 (T 500) and the literal net (T 1000).
    python tools/beamtime.py"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from neuralasr_amd.engine import Engine
for wl in ('bilstm3x500','literal'):
    spec, name = bench.workload_spec(wl)
    eng = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes, learning_rate=1e-4)
    eng.set_params(bench.init_params(eng.tensors(), seed=1))
    feats, seq_len, labels, label_len = bench.synth_batch(spec, 16, 500, seed=1234)
    logits = eng.forward(feats, seq_len)
    # a few training steps make the distribution peaky (blank-dominated), like a real run
    for peak in (False, True):
        if peak:
            for _ in range(60): eng.train_step(feats, seq_len, labels, label_len)
            logits = eng.forward(feats, seq_len)
        sl = [int(s) * (2 if wl=='literal' else 1) for s in seq_len] if False else seq_len
        t0=time.perf_counter(); n=3
        for _ in range(n): hy,_ = eng.beam_search(logits, seq_len, 100)
        dt=(time.perf_counter()-t0)/n*1e3
        t0=time.perf_counter()
        for _ in range(n): hy1,_ = eng.beam_search(logits[:, :1], seq_len[:1], 100)
        d1=(time.perf_counter()-t0)/n*1e3
        print(wl, 'logit frames', logits.shape[0], 'after 60 steps' if peak else 'random init', 'beam_ms(B16)=%.1f'%dt, 'one utterance %.1f ms'%d1, 'mean hyp len %.1f'%np.mean([len(h) for h in hy]), flush=True)
    eng.close()
