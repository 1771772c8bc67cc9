import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from neuralasr_amd.engine import Engine
from neuralasr_amd.utils import include_context
spec, _ = bench.workload_spec('bilstm3x500')
eng = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes, learning_rate=1e-4)
eng.set_params(bench.init_params(eng.tensors(), seed=1))
B,T=16,500
rs=np.random.RandomState(1)
_, seq_len, labels, label_len = bench.synth_batch(spec, B, T, seed=3)
fctx = np.zeros((B,T,546),np.float32)
for b in range(B): fctx[b]=include_context(rs.randn(T,26).astype(np.float32),10,26)
big = np.random.randn(B,T,546).astype(np.float32)
def run(name, decode, results, settle, hostwork, n=40):
    eng.set_step_decode(decode)
    t = eng.stage_batch(fctx, seq_len, labels, label_len, 10, 26)
    tok_prev=None
    def step(t):
        eng.commit_batch(t); eng.compute_grads(); eng.apply_adam(1.0); return eng.step_token()
    for i in range(5):
        tok=step(t); t = eng.stage_batch(fctx, seq_len, labels, label_len, 10, 26)
    eng.synchronize(); t0=time.perf_counter()
    for i in range(n):
        tok = step(t)
        if hostwork:
            x = big.copy(); y = np.ascontiguousarray(x[:, :, 260:286]); del x
        t = eng.stage_batch(fctx, seq_len, labels, label_len, 10, 26)
        if results: eng.step_results(B, T)
        if settle and tok_prev: eng.settle_token(tok_prev)
        tok_prev = tok
    eng.synchronize(); dt=(time.perf_counter()-t0)/n*1e3
    eng.discard_batch(t)
    print('%-60s %.3f ms/step'%(name,dt), flush=True)
for rep in range(2):
    run('staged loop', False, False, False, False)
    run('+ step decode', True, False, False, False)
    run('+ step_results wait', True, True, False, False)
    run('+ settle_token(prev)', True, True, True, False)
    run('+ host memory work (35 MB of copies)', True, True, True, True)
eng.close()
