#!/usr/bin/env python3
"""End-to-end cost of the training LOOP (neuralasr_amd.train.train_model: pickled utterances -> DataSet -> Network.train,
i.e. the reference's train.py:14-47 with its own data path) against the resident-batch step bench.py times.

Writes a synthetic .scp / .pkl set (26 MFCC x 21 context = 546 features, 500 frames, labels 40..80, C = 29) to a scratch
directory, trains the 3x500 bidirectional net for a few epochs at batch 16 on one GPU and prints the wall time per step of
the last epochs - with the loader thread staging every next batch (the default) and with synchronous uploads.

    python tools/e2e_train.py [utterances=64] [epochs=4] [network=networks.bilstm_ctc_net.BiLstm3x500CTCNet] [batch=16]
    python tools/e2e_train.py 128 4 networks.deepspeech.DeepSpeech 32      # BASELINE.json configs[3] per GPU"""
import os
import pickle
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)

from neuralasr_amd import train as train_mod          # noqa: E402
from neuralasr_amd.audiosample import AudioSample      # noqa: E402
from neuralasr_amd.config import Config               # noqa: E402
from neuralasr_amd.dataset import DataSet              # noqa: E402
from neuralasr_amd.symbols import Symbols              # noqa: E402
from neuralasr_amd.utils import include_context       # noqa: E402

CONFIG = """[Parameters]
samplerate=16000
numcep=26
numcontext=10
label_context=0
batch_size=%(batch)d
epochs=%(epochs)d
learningrate=0.0001
model_dir=%(out)s/model
start_step=0
report_step=1000000
num_gpus=1
punc_regex=[^a-z0-9 ]
sym_file=${MFCC Featurizer:output}/symbols
network=%(network)s

[Train]
input=${MFCC Featurizer:output}/train.scp

[Test]

[MFCC Featurizer]
input=unused.csv
output=%(out)s
"""


def main():
    n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    network = sys.argv[3] if len(sys.argv) > 3 else 'networks.bilstm_ctc_net.BiLstm3x500CTCNet'
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    out = tempfile.mkdtemp(prefix='nasr_e2e_')
    try:
        sym = Symbols(0)
        sym.insert_padding()
        for c in 'abcdefghijklmnopqrstuvwxyz_':
            sym.insert_sym(c)
        sym.insert_blank()
        sym.write(os.path.join(out, 'symbols'))
        rs = np.random.RandomState(7)
        names = []
        for i in range(n_utt):
            st = include_context(rs.randn(500, 26).astype(np.float32), 10, 26)
            st = ((st - st.mean()) / st.std()).astype(np.float32)          # utils.py:29
            lab = rs.randint(1, 28, size=rs.randint(40, 81)).astype(np.int32)
            with open(os.path.join(out, 'u%d.pkl' % i), 'wb') as fh:
                pickle.dump(AudioSample('u%d' % i, st, lab, ''), fh, pickle.HIGHEST_PROTOCOL)      # preprocess_mfcc.py:60
            names.append('u%d.pkl' % i)
        with open(os.path.join(out, 'train.scp'), 'w') as fh:
            fh.write('\n'.join(names) + '\n')
        cfgp = os.path.join(out, 'e2e.config')
        with open(cfgp, 'w') as fh:
            fh.write(CONFIG % dict(out=out, epochs=epochs, network=network, batch=batch))
        steps_per_epoch = (n_utt + batch - 1) // batch
        from neuralasr_amd.networks.hipnetwork import HipNetwork
        for label, prefetch, decoder in (
                ('overlapped loop (next batch loaded + staged under the running step), mean_ler by beam search on host threads', 2, 'beam'),
                ('the same with the greedy decoder on the device', 2, 'greedy'),
                ('reference order: load, upload, train (beam search waited for in every step)', 0, 'beam'),
                ('overlapped loop, beam, again', 2, 'beam'),
                ('overlapped loop, greedy, again', 2, 'greedy')):
            HipNetwork.train_ler_decoder = decoder
            cfg = Config(cfgp, True)
            stamps = []
            net_cls = cfg.load_network.__func__

            def load(self, fortraining=False, _orig=net_cls):
                net = _orig(self, fortraining)
                for name in ('train', 'finish_step'):
                    inner = getattr(net, name)

                    def timed(*a, _inner=inner, _name=name, **kw):
                        r = _inner(*a, **kw)
                        if _name == 'finish_step':
                            stamps.append(time.perf_counter())
                        return r
                    setattr(net, name, timed)
                return net
            cfg.load_network = load.__get__(cfg)
            train_mod.train_model(DataSet(cfg.train_input, cfg), None, cfg, prefetch=prefetch)
            stamps = sorted(set(stamps))
            warm = steps_per_epoch                           # the first epoch pays allocations and graph captures
            dt = (stamps[-1] - stamps[warm]) / (len(stamps) - 1 - warm)
            gaps = np.diff(np.asarray(stamps)) * 1e3
            med = float(np.median(gaps[warm:]))
            print('%-112s mean %.3f ms, median %.3f ms per step  (%d steps of %d frames: %.3f M frames/s at the median)'
                  % (label, dt * 1e3, med, len(stamps) - 1 - warm, 500 * batch, 500 * batch / med / 1e3), flush=True)
            print('   step-to-step gaps (ms): ' + ' '.join('%.1f' % g for g in gaps), flush=True)
        if os.environ.get('NASR_E2E_TIMERS'):
            # wall time of the engine calls train() and the loader thread make (monkeypatched timers), staged loop
            from neuralasr_amd.engine import Engine
            acc = {}

            def wrap(name):
                orig = getattr(Engine, name)

                def timed(self, *a, **k):
                    t0 = time.perf_counter()
                    try:
                        return orig(self, *a, **k)
                    finally:
                        d = acc.setdefault(name, [0.0, 0])
                        d[0] += time.perf_counter() - t0
                        d[1] += 1
                setattr(Engine, name, timed)
                return orig
            names = ['stage_batch', 'commit_batch', 'compute_grads', 'apply_adam', 'step_results', 'settle_step', 'settle_token', 'step_logits', 'beam_search', 'set_step_decode',
                     'label_error_rate', 'upload_batch_context', 'upload_batch']
            origs = {nm: wrap(nm) for nm in names if hasattr(Engine, nm)}
            gnb = DataSet.get_next_batch

            def timed_gnb(self):
                t0 = time.perf_counter()
                try:
                    return gnb(self)
                finally:
                    d = acc.setdefault('DataSet.get_next_batch', [0.0, 0])
                    d[0] += time.perf_counter() - t0
                    d[1] += 1
            DataSet.get_next_batch = timed_gnb
            for label, prefetch in (('overlapped loop', 2), ('reference order', 0)):
                acc.clear()
                cfg = Config(cfgp, True)
                train_mod.train_model(DataSet(cfg.train_input, cfg), None, cfg, prefetch=prefetch)
                print(' ' + label)
                for nm, (tot, cnt) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
                    print('   %-22s %4d calls  %8.3f ms each' % (nm, cnt, 1e3 * tot / cnt))
            for nm, o in origs.items():
                setattr(Engine, nm, o)
        if os.environ.get('NASR_E2E_PROFILE'):
            import cProfile
            import pstats
            cfg = Config(cfgp, True)
            pr = cProfile.Profile()
            pr.enable()
            train_mod.train_model(DataSet(cfg.train_input, cfg), None, cfg, prefetch=2)
            pr.disable()
            pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
    finally:
        shutil.rmtree(out, ignore_errors=True)


if __name__ == '__main__':
    main()
