import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
if len(sys.argv) > 1:
    from oracle import nasr_oracle as O
    from neuralasr_amd.engine import Engine
    spec = O.ModelSpec(546, 256, 1, True, 'concat', 29, pre=(256, 256, 512), post=256, relu_clip=20.0, dropout=(0.05,) * 4)
    B, T = 8, 60
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=3 * B + T, var_len=True, Lmin=1, Lmax=12)
    rs = np.random.RandomState(4)
    params = [p + 0.15 * rs.randn(*p.shape) for p in O.init_params(spec, seed=4)]
    e = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
               pre=spec.pre, post=spec.post, relu_clip=spec.relu_clip, dropout=spec.dropout)
    e.set_params(O.flatten(params))
    e.set_dropout_state(4567, 11)
    logits = e.forward(feats, seq_len)
    np.save(sys.argv[1] + '.logits.npy', logits)
    e.set_dropout_state(4567, 11)
    loss, nll, grads = e.loss_and_grads(feats, seq_len, labels, label_len)
    np.save(sys.argv[1], grads)
    json.dump([[n, o, r, c] for n, o, r, c in e.tensors()], open(sys.argv[1] + '.json', 'w'))
else:
    for mode in ('tp3', 'tp', 'tp_b'):
        env = dict(os.environ, NASR_GEMM=mode[:3].rstrip('_'))
        subprocess.check_call([sys.executable, __file__, f'/tmp/g_{mode}.npy'], env=env)
    a, b = np.load('/tmp/g_tp3.npy'), np.load('/tmp/g_tp.npy')
    c2 = np.load('/tmp/g_tp_b.npy')
    print('tph run-to-run max abs diff', np.abs(b - c2).max())
    from oracle import nasr_oracle as O
    spec = O.ModelSpec(546, 256, 1, True, 'concat', 29, pre=(256, 256, 512), post=256, relu_clip=20.0, dropout=(0.05,) * 4)
    B, T = 8, 60
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=3 * B + T, var_len=True, Lmin=1, Lmax=12)
    rs = np.random.RandomState(4)
    params = [p + 0.15 * rs.randn(*p.shape) for p in O.init_params(spec, seed=4)]
    loss_o, nll_o, grads_o, logits_o = O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len, drop=(4567, 11))
    go = O.flatten(grads_o)
    for mode in ('tp3', 'tp'):
        lg = np.load(f'/tmp/g_{mode}.npy.logits.npy')
        print(mode, 'logits max abs err vs oracle', float(np.abs(lg - logits_o).max()), 'rel L2', float(np.linalg.norm(lg - logits_o) / np.linalg.norm(logits_o)))
    for (n, o, r, c), g_o in zip(json.load(open('/tmp/g_tp.npy.json')), grads_o):
        x, y, z = a[o:o + r * c], b[o:o + r * c], np.asarray(g_o).ravel()
        nz = np.linalg.norm(z) + 1e-30
        if n in ('b5', 'h5'):
            d = np.abs(y - z)
            k = np.argsort(-d)[:6]
            print(n, 'largest errors at', k.tolist(), 'tph', y[k].tolist(), 'oracle', z[k].tolist(), 'median err', float(np.median(d)))
        print(f'{n:12s} {r:5d}x{c:5d} tp3-vs-tph {np.linalg.norm(x - y) / nz:.2e}  tp3-vs-oracle {np.linalg.norm(x - z) / nz:.2e}  tph-vs-oracle {np.linalg.norm(y - z) / nz:.2e}')
