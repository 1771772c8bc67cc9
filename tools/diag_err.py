import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import nasr_oracle as O
from neuralasr_amd.engine import Engine
def rel(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
for merge in ('stack_reshape', 'concat'):
  for T, var in ((100, True), (250, True), (500, False), (500, True)):
    spec = O.ModelSpec(546, 500, 1, True, merge, 29)
    feats, seq_len, labels, label_len = O.synth_batch(spec, 16, T, seed=1234, var_len=var)
    params = [p.astype(np.float32).astype(np.float64) for p in O.init_params(spec, seed=1)]
    e = Engine(546, 500, 1, True, merge, 29)
    e.set_params(O.flatten(params))
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    lo, nllo, go, _ = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    errs = []
    for (name, off, r, c), g_o in zip(e.tensors(), go):
        errs.append('%s %.1e' % (name.split('/')[-2][:2] + name[-1] if '/' in name else name, rel(grads[off:off+r*c].reshape(g_o.shape), g_o)))
    print(merge, 'T', T, 'var', var, 'loss rel %.1e' % (abs(loss-lo)/lo), 'nll max rel %.1e' % np.max(np.abs(nll-nllo)/nllo), ' '.join(errs), flush=True)
    e.close()
