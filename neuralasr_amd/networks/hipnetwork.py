"""HipNetwork — the counterpart of TensorFlowNetwork (reference: networks/tfnetwork.py:13-190): numpy-in /
numpy-out train / validate / evaluate / decode, checkpoints, and data-parallel training, with every
arithmetic step in the HIP library behind include/nasr.h (no TensorFlow, no CPU fallback).

Differences from the reference that a caller can observe, all deliberate and documented in DESIGN.md:
  * train() / validate() / evaluate() / decode() all use the reference's decoder (ctc_beam_search_decoder, width 100,
    merge_repeated, host C++).  The `mean_ler` of a training step is decoded from the step's own logits, copied out
    behind the CTC kernels, while the device runs the backward pass; train_model does not even wait for it - it
    collects the steps' LERs when it logs (`train_ler_decoder = 'greedy'` switches to the greedy decoder the
    reference names in the comment at tfnetwork.py:62-63: no host work at all);
  * num_gpus > 1 means one process per GPU (torch.distributed.run); inside a single process the towers are
    time-sliced on one GPU with the same split / averaging arithmetic;
  * checkpoints are `model-<step>.npz` (flat fp32 params + Adam m, v + step) with TF Saver's cadence and
    keep-5 policy, not TF's format."""
import glob
import json
import os
import shutil
import threading
import weakref

import numpy as np

from ..engine import Engine
from ..parallel import Collective, take_shard
from .network import Network


def _glorot_init(tensors, seed):
    """Default TF initialisers the reference relies on: glorot-uniform LSTM kernels, zero biases,
    xavier-normal W (networks/bilstm_ctc_net.py:35-36), zero b.  TF's own seeded stream
    (tf.set_random_seed(1), tfnetwork.py:19) is not reproducible outside TF."""
    rs = np.random.RandomState(seed)
    chunks = []
    for name, _, rows, cols in tensors:
        if name.endswith('kernel'):
            lim = np.sqrt(6.0 / (rows + cols))
            chunks.append(rs.uniform(-lim, lim, size=rows * cols))
        elif name == 'W':
            chunks.append(rs.randn(rows * cols) * np.sqrt(2.0 / (rows + cols)))
        else:
            chunks.append(np.zeros(rows * cols))
    return np.concatenate(chunks).astype(np.float32)


class HipNetwork(Network):
    # model shape; subclasses override (the reference hard-codes these as locals of create_network)
    num_hidden = 500
    num_layers = 1
    bidirectional = True
    merge = 'stack_reshape'
    keep_checkpoints = 5                     # tf.train.Saver() default max_to_keep
    decoder = 'beam'                         # validate / evaluate / decode: tf.nn.ctc_beam_search_decoder defaults
    train_ler_decoder = 'beam'               # mean_ler of train(): the reference's decoder on the step's own logits ('greedy': on the device)
    beam_width = 100
    device_context = True                    # rebuild include_context's stacking on the GPU (1/(2c+1) of the H2D bytes)
    bucketed_allreduce = True                # one process per GPU: exchange per-layer gradient buckets under the backward pass
    # train() returns as soon as the step's loss and decode exist (after the forward pass + CTC), while the device runs the
    # backward pass, the exchange and Adam: the next step is enqueued behind it and the device never waits for the host.
    # A void step (rare: a persistent-recurrence abort on some rank) is noticed one call later and repeated then.
    # False: train() additionally waits for the END of its step (and repeats it there if it was void) - same code path.
    async_step = True

    def __init__(self, config, fortraining=False):
        Network.__init__(self)
        self.fortraining = fortraining
        self.config = config
        self.num_classes = config.symbols.counter
        self.coll = Collective()
        device = int(os.environ.get('LOCAL_RANK', '0')) if self.coll.world > 1 else 0
        stream = None
        if self.coll.world > 1:
            # a dedicated non-default torch stream, made current: the engine launches on it and the RCCL
            # all-reduce is ordered against it (the legacy default stream is neither capturable nor shared)
            import torch
            torch.cuda.set_device(device)
            self._torch_stream = torch.cuda.Stream(device=device)
            torch.cuda.set_stream(self._torch_stream)
            stream = self._torch_stream.cuda_stream
        self.logger.info('Initializing network for %s.' % ('training' if fortraining else 'inference'))
        self.engine = self.make_engine(config, device, stream)
        self.engine.set_step_decode(True)
        self.engine.set_params(self.initial_params(self.engine.tensors(), seed=1))
        self._grad_tensor = None
        self._reducer = None
        self._staged = {}                       # id(mfccs) -> (ticket, weakref to mfccs): batches stage_batch() sent ahead
        self._staged_lock = threading.Lock()
        self._pending = None                    # the batch of the last fast-path step, until that step is known not to be void
        self._begun = None                      # begin_step() without its finish_step() yet
        self._pool = None                       # host threads that decode the steps' logits (train_ler_decoder = 'beam')
        self.global_step = self.config.start_step
        self.load_checkpoint(self.global_step if fortraining else 1, self.config.model_dir)
        if fortraining and self.coll.rank == 0:
            self.write_config()
            self.config.symbols.write(os.path.join(self.config.model_dir, os.path.basename(self.config.sym_file)))

    # the two hooks a model family overrides (DeepSpeech does): how the engine is shaped, how variables start
    def make_engine(self, config, device, stream):
        return Engine(config.feature_size, self.num_hidden, self.num_layers, self.bidirectional, self.merge,
                      self.num_classes, learning_rate=config.learningrate, device_id=device, stream=stream)

    def initial_params(self, tensors, seed):
        return _glorot_init(tensors, seed)

    # ------------------------------------------------------------------ per-model hook
    def create_network(self, features, labels, seq_len, labels_len, num_classes, is_training):
        """(logits [T',B,C], loss, decoded ids per utterance, None, mean LER) for a batch
        (reference: networks/bilstm_ctc_net.py:10-52 returns the same five graph nodes)."""
        logits = self.engine.forward(features, seq_len)
        loss, _ = self.engine.loss(features, seq_len, labels, labels_len)
        hyps = self.engine.get_decoded(len(seq_len), np.asarray(features).shape[1])
        ler = self.engine.label_error_rate(hyps, labels, labels_len)
        return logits, loss, hyps, None, ler

    # ------------------------------------------------------------------ checkpoints
    def _ckpt_files(self, model_dir):
        files = glob.glob(os.path.join(model_dir, 'model-*.npz'))
        return sorted(files, key=lambda f: int(os.path.basename(f)[6:-4]))

    def load_checkpoint(self, start_epoch, model_dir):
        if start_epoch > 0:
            self.logger.info('Restoring checkpoint: ' + model_dir)
            files = self._ckpt_files(model_dir)
            if not files:
                raise FileNotFoundError('no checkpoint (model-<step>.npz) in ' + model_dir)
            with np.load(files[-1]) as z:
                self.engine.set_params(z['params'])
                self.engine.set_adam_state(z['adam_m'], z['adam_v'], int(z['step']))
            self.logger.info('Done Restoring checkpoint: ' + files[-1])
        elif self.coll.rank == 0:
            if os.path.exists(model_dir):
                shutil.rmtree(model_dir)
            os.makedirs(model_dir)

    def write_config(self):
        dst = os.path.join(self.config.model_dir, os.path.basename(self.config.configfile))
        if os.path.exists(dst):
            self.logger.warning('Not overwriting. Config file already exists: ' + dst)
        else:
            self.config.write(dst)

    def save_checkpoint(self):
        self._settle()
        if self.coll.rank != 0:
            return
        m, v, step = self.engine.get_adam_state()
        path = os.path.join(self.config.model_dir, 'model-%d.npz' % self.global_step)
        np.savez(path, params=self.engine.get_params(), adam_m=m, adam_v=v, step=np.int64(step),
                 meta=json.dumps({'hidden': self.num_hidden, 'layers': self.num_layers,
                                  'bidirectional': self.bidirectional, 'merge': self.merge,
                                  'classes': self.num_classes, 'feature_size': self.config.feature_size}))
        for old in self._ckpt_files(self.config.model_dir)[:-self.keep_checkpoints]:
            os.remove(old)

    # ------------------------------------------------------------------ numpy-in / numpy-out API
    def _towers(self):
        """(n, owned tower indices): one tower per process under torch.distributed, otherwise all
        num_gpus towers time-sliced in this process."""
        n = max(1, int(self.config.num_gpus)) if self.fortraining else 1
        if self.coll.world > 1:
            if n != self.coll.world:
                raise ValueError('config num_gpus=%d but %d processes were launched' % (n, self.coll.world))
            return n, [self.coll.rank]
        return n, list(range(n))

    @staticmethod
    def _is_abort(exc):
        return 'persistent recurrence aborted' in str(exc) or 'training step is void' in str(exc)

    def _retry_aborted(self, fn):
        """Run fn(); if this rank's persistent recurrence gave up in it (the handle has then switched to the per-step
        kernels, nasr_api.hip persist_check), run it once more.  Forward-only calls have no collective inside, so a
        local repeat keeps multi-rank runs in step."""
        from .._lib import NasrError
        try:
            return fn()
        except NasrError as exc:
            if not self._is_abort(exc):
                raise
            self.logger.warning('persistent recurrence aborted in a forward-only call: repeating it on the per-step kernels')
            return fn()

    def _decode(self, mfccs, seq_len, which):
        if which == 'beam':
            logits = self._retry_aborted(lambda: self.engine.forward(mfccs, seq_len))
            return self.engine.beam_search(logits, seq_len, self.beam_width, merge_repeated=True)[0]
        return self._retry_aborted(lambda: self.engine.greedy_decode(mfccs, seq_len))

    def _loss_ler_one(self, mfccs, labels, seq_len, labels_len):
        def run():
            loss, _ = self.engine.loss(mfccs, seq_len, labels, labels_len)
            hyps = None if self.decoder == 'beam' else self.engine.get_decoded(len(seq_len), np.asarray(mfccs).shape[1])
            return loss, hyps
        loss, hyps = self._retry_aborted(run)
        if hyps is None:
            hyps = self._decode(mfccs, seq_len, 'beam')
        return loss, self.engine.label_error_rate(hyps, labels, labels_len), hyps

    def _loss_ler(self, mfccs, labels, seq_len, labels_len):
        """(loss, mean LER, hypotheses).  A training network evaluates the way its graph was built
        (setup_training_network, tfnetwork.py:115-140): per tower on the tf.split shards - the literal net's
        stack-reshape map is a function of the SHARD's batch size - and the mean of the shard means."""
        n, mine = self._towers()
        if n == 1:
            return self._loss_ler_one(mfccs, labels, seq_len, labels_len)
        losses, lers, hyps = [], [], []
        for k in mine:
            f, l, s, ll = take_shard(mfccs, labels, seq_len, labels_len, n, k)
            lo, le, hy = self._loss_ler_one(f, l, s, ll)
            losses.append(lo)
            lers.append(le)
            hyps.extend(hy)
        loss, ler = float(np.mean(losses)), float(np.mean(lers))
        if self.coll.world > 1:
            loss, ler = self.coll.mean_scalars([loss, ler])
        return loss, ler, hyps

    def train(self, mfccs, labels, seq_len, labels_len):
        self.begin_step(mfccs, labels, seq_len, labels_len)
        return self.finish_step()

    def begin_step(self, mfccs, labels, seq_len, labels_len):
        """First half of train(): everything of the optimisation step is ENQUEUED (upload or commit of the staged batch,
        forward, CTC, backward, gradient exchange, Adam) and the call returns.  The host is free until finish_step() -
        train_model loads and stages the next batch there, under the step the device is running (the reference loads the
        next batch inside the timed step with the device idle, train.py:23-26)."""
        self.global_step += 1
        n, mine = self._towers()
        if len(mine) == 1:
            f, l, s, ll = take_shard(mfccs, labels, seq_len, labels_len, n, mine[0])
            batch = (mfccs, f, l, s, ll, n)
            self._begun = ('one', (batch, self._enqueue_step(*batch)))
        else:
            self._begun = ('sliced', (mfccs, labels, seq_len, labels_len))

    def finish_step(self, lazy=False):
        """Second half of train(): (loss, mean_ler) of the step begun last - available after its forward pass + CTC; the
        device may still be in its backward pass when this returns (async_step = False: it has ended, and was repeated if it
        was void).  lazy=True: mean_ler may come back as a handle with a .result() (the beam search of the step's logits
        still runs on host threads) - for a caller like train_model, which only needs the LERs when it logs; multi-process
        runs average such a window with mean_over_ranks()."""
        kind, batch = self._begun
        self._begun = None
        if kind == 'sliced':
            return self._train_sliced(*batch)
        out = self._finish_async(*batch, lazy=lazy)
        if not self.async_step:
            self._settle()
        return out

    def mean_over_ranks(self, value):
        """Mean of a host float over the towers' processes (reduce_mean of tfnetwork.py:135-136); the value itself with one."""
        return self.coll.mean_scalars([value])[0] if self.coll.world > 1 else float(value)

    # ------------------------------------------------------------------ the fast path of train()
    def _enqueue_step(self, mfccs, f, l, s, ll, n):
        """Everything of one optimisation step, enqueued without waiting for any of it.  Returns the step's token
        (nasr_step_token): the name under which its end can be asked about later, however many steps follow."""
        ticket = self._take_staged(mfccs)
        if ticket is not None:
            self.engine.commit_batch(ticket)
        elif not (self._use_device_context() and
                  self.engine.upload_batch_context(f, s, l, ll, self.config.numcontext, self.config.numcep)):
            self.engine.upload_batch(f, s, l, ll)
        beam = self.train_ler_decoder == 'beam'
        self.engine.set_step_decode(True, logits=beam, greedy=not beam)     # (the beam search reads the logits; no greedy pass then)
        self.engine.compute_grads()
        if self.coll.world > 1:
            if self._grad_tensor is None:
                self._grad_tensor = self.engine.grad_tensor()
                self._reducer = self.coll.bucketed(self.engine, self._grad_tensor) if self.bucketed_allreduce else None
            if self._reducer is not None:
                self._reducer.all_reduce()
            else:
                self.coll.all_reduce_sum_(self._grad_tensor)
        self.engine.apply_adam(1.0 / n)
        return self.engine.step_token()

    # ------------------------------------------------------------------ the step's LER (tfnetwork.py:61-70)
    def _beam_ler(self, logits, s, l, ll):
        hyps = self.engine.beam_search(logits, s, self.beam_width, merge_repeated=True)[0]
        return self.engine.label_error_rate(hyps, l, ll)

    def _step_ler(self, hyps, f, l, s, ll, lazy):
        """mean_ler of the step just enqueued: greedy from the device's decode, or the reference's beam search on the step's
        own logits - on a host thread (the C++ decoder runs one thread per utterance and holds no interpreter lock)."""
        if self.train_ler_decoder != 'beam':
            return self.engine.label_error_rate(hyps, l, ll)
        logits = self.engine.step_logits(len(s), f.shape[1])
        if not lazy:
            return self._beam_ler(logits, s, l, ll)
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=4, thread_name_prefix='nasr-beam')
        return self._pool.submit(self._beam_ler, logits, s, l, ll)

    def _redo_step(self, batch, why):
        """A void step's batch again, start to finish, until it counts; returns the (loss, mean_ler) of the attempt that
        did.  Every rank takes this path at the same point of its call sequence and runs the same number of attempts: the
        fault word is all-reduced with the gradients, so 'void' is the same answer everywhere."""
        mfccs, f, l, s, ll, n = batch
        for attempt in range(3):
            self.logger.warning('%s: repeating its batch' % why)
            token = self._enqueue_step(None, f, l, s, ll, n)
            loss, fault, hyps = self.engine.step_results(len(s), f.shape[1])
            ler = float('nan') if fault else self._step_ler(hyps, f, l, s, ll, lazy=False)
            if not self.engine.settle_token(token):
                return loss, ler
        raise RuntimeError('a training step stayed void after 3 attempts')

    def _settle(self):
        """Before anything that is not the next fast-path step (validation, checkpoint, decode, the slow path): wait for
        the last fast-path step and repeat it if it was void."""
        if self._pending is not None:
            (batch, token), self._pending = self._pending, None
            if self.engine.settle_token(token):
                self._redo_step(batch, 'the last training step was void (persistent recurrence aborted on some rank)')

    def _finish_async(self, batch, token, lazy=False):
        mfccs, f, l, s, ll, n = batch
        loss, fwd_fault, hyps = self.engine.step_results(len(s), f.shape[1])
        if fwd_fault:
            # THIS rank's forward recurrence aborted: its loss and decode mean nothing.  The step is void on every rank
            # (the fault word travels with the gradients); it is repeated where every rank notices it.
            loss, ler = float('nan'), float('nan')
        else:
            ler = self._step_ler(hyps, f, l, s, ll, lazy)
        # the step BEFORE this one has ended by now (stream order): was it void?
        prev, self._pending = self._pending, (batch, token)
        if prev is not None and self.engine.settle_token(prev[1]):
            # This step was enqueued before anybody knew: whatever voided the previous one (a co-tenant, a placement) may
            # have voided it too.  Learn its fate BEFORE anything else is enqueued, then repeat what was void, in order.
            self._pending = None
            cur_void = self.engine.settle_token(token)
            self._redo_step(prev[0], 'step %d was void (persistent recurrence aborted on some rank)' % (self.global_step - 1))
            if cur_void:
                loss, ler = self._redo_step(batch, 'step %d was void as well' % self.global_step)
        elif fwd_fault and self.coll.world == 1:
            # single process: nobody else to keep in step with - settle this one now and report the repeat's values
            self._pending = None
            self.engine.settle_token(token)
            loss, ler = self._redo_step(batch, 'step %d was void (persistent recurrence aborted)' % self.global_step)
        if hasattr(ler, 'result'):
            # lazy: the LER is still being decoded; the caller averages its window over the ranks (mean_over_ranks)
            if self.coll.world > 1:
                loss = self.coll.mean_scalars([loss])[0]
            return np.float32(loss), ler
        if self.coll.world > 1:
            # (a rank-local forward fault leaves NaN here on every rank: train_model keeps such a step out of its means;
            # the step itself is repeated one call later, where every rank sees the fault word)
            loss, ler = self.coll.mean_scalars([loss, ler])
        return np.float32(loss), np.float32(ler)

    def _use_device_context(self):
        # rand_shift's roll-and-crop (dataset.py:23-31) leaves real neighbour frames where include_context put its pad
        # in the first / last numcontext frames: such batches are uploaded whole
        return (self.device_context and getattr(self.config, 'numcontext', 0) > 0 and
                not getattr(self.config, 'rand_shift', 0) > 0)

    # ------------------------------------------------------------------ input pipeline
    def stage_batch(self, mfccs, labels, seq_len, labels_len):
        """Send a batch that train() will be given NEXT towards the GPU now: this process's shard goes through pinned
        memory and the engine's copy stream while the current step computes (the reference loads and feeds the next
        batch inside the timed step, dataset.py:33-40, train.py:23-26).  Safe to call from DataSet.prefetch's loader
        thread.  train() recognises the batch by identity; anything not staged is uploaded the synchronous way."""
        n, mine = self._towers()
        if len(mine) != 1:
            return False                        # towers time-sliced on one GPU: each upload replaces the resident batch
        f, l, s, ll = take_shard(mfccs, labels, seq_len, labels_len, n, mine[0])
        ctx = self.config.numcontext if self._use_device_context() else 0
        ticket = self.engine.stage_batch(f, s, l, ll, ctx, getattr(self.config, 'numcep', 0))
        if ticket is None:
            return False
        with self._staged_lock:
            self._staged[id(mfccs)] = (ticket, weakref.ref(mfccs))
        return True

    def _take_staged(self, mfccs):
        with self._staged_lock:
            ent = self._staged.pop(id(mfccs), None)
        if ent is None:
            return None
        if ent[1]() is not mfccs:               # the id was recycled by another array: the staged batch is an orphan
            self.engine.discard_batch(ent[0])
            return None
        return ent[0]

    def discard_staged(self):
        """Give back the slots of batches that were staged but never trained on (the loop ended or raised)."""
        self._settle()
        with self._staged_lock:
            ents, self._staged = list(self._staged.values()), {}
        for ticket, _ in ents:
            try:
                self.engine.discard_batch(ticket)
            except Exception:                   # noqa: BLE001 - already committed or discarded
                pass

    def _train_sliced(self, mfccs, labels, seq_len, labels_len):
        """num_gpus towers time-sliced on the one GPU of a single process: each tower's shard in turn through the engine, the
        gradients summed on the host (average_gradients, tfnetwork.py:72-86), one Adam step; a void tower (its persistent
        recurrence gave up: the handle runs the per-step kernels from then on) repeats the whole step."""
        from .._lib import NasrError
        self._settle()
        n, mine = self._towers()
        for attempt in range(3):
            losses, lers, gsum, void = [], [], None, False
            for k in mine:
                f, l, s, ll = take_shard(mfccs, labels, seq_len, labels_len, n, k)
                if not (self._use_device_context() and
                        self.engine.upload_batch_context(f, s, l, ll, self.config.numcontext, self.config.numcep)):
                    self.engine.upload_batch(f, s, l, ll)
                beam = self.train_ler_decoder == 'beam'
                self.engine.set_step_decode(True, logits=beam, greedy=not beam)
                self.engine.compute_grads()
                try:
                    losses.append(self.engine.get_loss())
                    hyps = None if self.train_ler_decoder == 'beam' else self.engine.get_decoded(len(s), f.shape[1])
                    lers.append(self._step_ler(hyps, f, l, s, ll, lazy=False))
                    g = self.engine.get_grads().astype(np.float64)
                except NasrError as exc:
                    if not self._is_abort(exc):
                        raise
                    void = True
                    break
                gsum = g if gsum is None else gsum + g
            if not void:
                self.engine.set_grads((gsum / n).astype(np.float32))
                self.engine.apply_adam(1.0)
                if not self.engine.step_void():
                    return np.float32(np.mean(losses)), np.float32(np.mean(lers))
            self.logger.warning('step %d was void (persistent recurrence aborted): repeating it' % self.global_step)
        raise RuntimeError('training step %d stayed void after 3 attempts' % self.global_step)

    def validate(self, mfccs, labels, seq_len, labels_len):
        self._settle()
        loss, ler, _ = self._loss_ler(mfccs, labels, seq_len, labels_len)
        return [np.float32(loss), np.float32(ler)]

    def evaluate(self, mfccs, labels, seq_len, labels_len):
        self._settle()
        loss, ler, hyps = self._loss_ler(mfccs, labels, seq_len, labels_len)
        # SparseTensorValue.values: every utterance's ids concatenated (tfnetwork.py:176-177)
        flat = np.asarray([i for h in hyps for i in h], dtype=np.int64)
        return flat, np.float32(loss), np.float32(ler)

    def decode(self, mfccs, seq_len):
        self._settle()
        hyps = self._decode(mfccs, seq_len, self.decoder)
        return np.asarray([i for h in hyps for i in h], dtype=np.int64)
