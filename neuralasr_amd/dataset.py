"""Batch reader over a .scp list of pickled AudioSample files (reference: dataset.py:12-91).
Same semantics: sequential batches of config.batch_size files, features and labels padded with the
padding id (0) to the batch maximum, the tail batch filled by re-loading the last file, optional
rand_shift roll-and-crop augmentation; returns (ndarray f32 [B,T,F], ndarray i32 [B,Lmax],
list of 0-d np.int32, list of int)."""
import io
import os
import pickle

import numpy as np

from .audiosample import AudioSample

_NUMPY_OK = {('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
             ('numpy', 'ndarray'), ('numpy', 'dtype'), ('numpy.core.multiarray', 'scalar'),
             ('numpy._core.multiarray', 'scalar'), ('numpy._core.numeric', '_frombuffer'),
             ('numpy.core.numeric', '_frombuffer'),
             # byte-string / plain-object reconstruction helpers pickle protocols 2-5 emit (no side effects)
             ('_codecs', 'encode'), ('copyreg', '_reconstructor'), ('copy_reg', '_reconstructor'),
             ('builtins', 'object'), ('__builtin__', 'object')}


class _SampleUnpickler(pickle.Unpickler):
    """Resolves `audiosample.AudioSample` (the module path preprocess_mfcc.py pickled) to our class and
    allows nothing beyond the NumPy array reconstructors: a .pkl is data, not code."""

    def find_class(self, module, name):
        if name == 'AudioSample' and module.split('.')[-1] == 'audiosample':
            return AudioSample
        if (module, name) in _NUMPY_OK:
            return super().find_class(module, name)
        raise pickle.UnpicklingError('refusing to load %s.%s from a sample file' % (module, name))


class _SwitchInterval:
    """Process-wide, reference-counted override of sys.setswitchinterval: the first generator in lowers it, the last one
    out restores what it found - nested or abandoned prefetch generators (train + validation) cannot leave each other's
    saved value behind."""

    def __init__(self, seconds):
        import threading
        self.seconds, self.users, self.saved, self.lock = seconds, 0, None, threading.Lock()

    def __enter__(self):
        import sys
        with self.lock:
            if self.users == 0:
                self.saved = sys.getswitchinterval()
                sys.setswitchinterval(min(self.saved, self.seconds))
            self.users += 1

    def __exit__(self, *exc):
        import sys
        with self.lock:
            self.users -= 1
            if self.users == 0:
                sys.setswitchinterval(self.saved)
        return False


_short_switch_interval = _SwitchInterval(2e-4)


class DataSet:
    def __init__(self, filename, config):
        self.filename = filename
        self.config = config
        self.padding_id = config.symbols.get_padding_id()
        self.index = 0
        root = os.path.dirname(self.filename)
        with open(self.filename, 'r') as fh:
            self.X = [os.path.join(root, line.strip()) for line in fh.readlines()]

    def augment_mfcc(self, mfcc):
        shift = self.config.rand_shift
        r = np.random.randint(-shift, shift)
        mfcc = np.roll(mfcc, r, axis=0)
        if r > 0:
            return mfcc[r:, :]
        if r < 0:
            return mfcc[:r, :]
        return mfcc

    def load_pkl(self, pklfilename):
        with open(pklfilename, 'rb') as fh:
            sample = _SampleUnpickler(io.BytesIO(fh.read())).load()
        if self.config.rand_shift > 0:
            sample.mfcc = self.augment_mfcc(sample.mfcc)
        return sample.mfcc, sample.labels, np.asarray(sample.mfcc.shape[0], dtype=np.int32), sample.labels.shape[0]

    def reset_epoch(self):
        self.index = 0

    def has_more_batches(self):
        return self.index < len(self.X)

    def get_next_batch(self):
        bs = self.config.batch_size
        items = [self.load_pkl(self.X[self.index])]
        self.index += 1
        while self.index % bs > 0:
            if self.index >= len(self.X):
                if len(items) == bs:
                    break
                self.index -= 1          # tail batch: keep re-loading the last file until it is full
            items.append(self.load_pkl(self.X[self.index]))
            self.index += 1
        max_time = max(m.shape[0] for m, _, _, _ in items)
        max_label = max(n for _, _, _, n in items)
        # pad with the padding id to the batch maximum (dataset.py:75-81: np.pad per utterance + np.asarray of the list);
        # written as ONE allocation filled in place - 17.5 MB per batch at 16 x 500 x 546 are copied once instead of twice
        pad = self.padding_id
        first_m, first_l = items[0][0], items[0][1]
        mfccs = np.empty((len(items), max_time) + first_m.shape[1:], dtype=np.result_type(*[m.dtype for m, _, _, _ in items]))
        labels = np.full((len(items), max_label) + first_l.shape[1:], pad, dtype=np.result_type(*[l.dtype for _, l, _, _ in items]))
        for i, (m, l, _, n) in enumerate(items):
            mfccs[i, :m.shape[0]] = m
            mfccs[i, m.shape[0]:] = pad         # only the tail is filled: no pass over the whole batch
            labels[i, :n] = l
        return mfccs, labels, [s for _, _, s, _ in items], [n for _, _, _, n in items]

    # ------------------------------------------------------------------ asynchronous input pipeline
    def prefetch(self, depth=2, stage=None):
        """Iterate the remaining batches of this epoch while a background thread unpickles and pads the next
        `depth` of them (the reference loads synchronously inside the timed loop, train.py:23-25).  Batch
        composition and order are exactly get_next_batch()'s; with rand_shift > 0 the augmentation draws from
        np.random in the loader thread, so the draws stay in batch order.  `stage` (optional): called with each
        batch's four arrays in the loader thread as soon as it is padded - HipNetwork.stage_batch starts its H2D copy
        there, under the step that is running (pinned staging + copy stream)."""
        import queue
        import threading
        q = queue.Queue(maxsize=max(1, depth))
        done = object()
        quit_ = threading.Event()                 # set when the consumer stops early (exception in a train step, close())

        def put(item):
            while not quit_.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def worker():
            try:
                while not quit_.is_set() and self.has_more_batches():
                    batch = self.get_next_batch()
                    if stage is not None:
                        stage(*batch)
                    if not put(batch):
                        return
                put(done)
            except BaseException as exc:      # surface loader errors in the consumer
                put(exc)

        # The loader thread holds the interpreter lock through unpickling and array assembly; with CPython's default
        # 5 ms switch interval the training thread can wait that long to get it back after the GPU step has finished.
        t = threading.Thread(target=worker, name='nasr-prefetch', daemon=True)
        with _short_switch_interval:
            t.start()
            try:
                while True:
                    item = q.get()
                    if item is done:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    yield item
            finally:
                quit_.set()
                t.join()

    def get_feature_shape(self):
        return [self.config.batch_size, None, self.config.feature_size]

    def get_label_shape(self):
        return [self.config.batch_size, None, 1]

    def get_num_of_sample(self):
        return len(self.X)
