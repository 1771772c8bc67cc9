"""Engine: NumPy-in / NumPy-out wrapper over one libnasr handle (= one GPU).  This is the thin layer the
`Network` plugin classes (neuralasr_amd/networks) and bench.py sit on; all arithmetic happens in the HIP
library behind include/nasr.h."""
import ctypes
from ctypes import POINTER, byref, c_char, c_float, c_int32, c_int64, c_uint32, c_void_p

import numpy as np

from . import _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _fp(a):
    return a.ctypes.data_as(POINTER(c_float))


def _ip(a):
    return a.ctypes.data_as(POINTER(c_int32))


class Engine:
    def __init__(self, feature_size, hidden, num_layers, bidirectional, merge, num_classes, forget_bias=1.0,
                 learning_rate=1e-4, beta1=0.9, beta2=0.999, epsilon=1e-8, device_id=0, stream=None,
                 pre=(), post=0, relu_clip=20.0, dropout=()):
        """`pre` / `post` / `relu_clip` / `dropout`: the clipped-ReLU dense stages of the DeepSpeech family
        (networks/deepspeech.py): widths of the stages in front of the LSTM stack, width of the one behind it (0 = none),
        and the drop probability of each of them in that order."""
        self.lib = _lib.load()
        merge_id = _lib.MERGE_BY_NAME[merge] if isinstance(merge, str) else int(merge)
        pre, dropout = tuple(int(w) for w in pre), tuple(float(p) for p in dropout)
        if len(pre) > 3 or len(dropout) > 4:
            raise ValueError('at most 3 dense stages before the LSTM stack and one behind it')
        self.cfg = _lib.ModelCfg(int(feature_size), int(hidden), int(num_layers), int(bool(bidirectional)), merge_id,
                                 int(num_classes), float(forget_bias), float(learning_rate), float(beta1),
                                 float(beta2), float(epsilon), len(pre), (c_int32 * 3)(*(pre + (0,) * (3 - len(pre)))),
                                 int(post), float(relu_clip),
                                 (c_float * 4)(*[dropout[i] if i < len(dropout) else 0.0 for i in range(4)]))
        if stream is not None and int(stream) == 0:
            raise ValueError('stream 0 (the legacy default stream) cannot carry the engine: pass a created stream '
                             '(e.g. torch.cuda.Stream().cuda_stream) or None for an engine-owned one')
        self.h = c_void_p()
        rc = self.lib.nasr_create(byref(self.cfg), int(device_id), c_void_p(stream) if stream else None, byref(self.h))
        if rc != 0:
            msg = self.lib.nasr_last_error(None)
            self.h = None
            raise _lib.NasrError(rc, msg.decode() if msg else 'nasr_create failed')
        self.num_classes = int(num_classes)
        self.param_count = int(self.lib.nasr_param_count(self.h))

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, 'h', None):
            self.lib.nasr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(self.lib, self.h, rc)

    @property
    def backend(self):
        return self.lib.nasr_backend(self.h).decode()

    def synchronize(self):
        self._ck(self.lib.nasr_synchronize(self.h))

    # ------------------------------------------------------------------ parameters
    def tensors(self):
        out = []
        for i in range(self.lib.nasr_num_tensors(self.h)):
            name = (c_char * 64)()
            off, r, c = c_int64(), c_int64(), c_int64()
            self._ck(self.lib.nasr_tensor_info(self.h, i, byref(name), byref(off), byref(r), byref(c)))
            out.append((name.value.decode(), off.value, r.value, c.value))
        return out

    def set_params(self, flat):
        flat = _f32(flat).ravel()
        self._ck(self.lib.nasr_set_params(self.h, _fp(flat), flat.size))

    def get_params(self):
        flat = np.empty(self.param_count, np.float32)
        self._ck(self.lib.nasr_get_params(self.h, _fp(flat), flat.size))
        return flat

    def set_adam_state(self, m, v, step):
        m, v = _f32(m).ravel(), _f32(v).ravel()
        self._ck(self.lib.nasr_set_adam_state(self.h, _fp(m), _fp(v), m.size, int(step)))

    def get_adam_state(self):
        m = np.empty(self.param_count, np.float32)
        v = np.empty(self.param_count, np.float32)
        step = c_int64()
        self._ck(self.lib.nasr_get_adam_state(self.h, _fp(m), _fp(v), m.size, byref(step)))
        return m, v, step.value

    def set_learning_rate(self, lr):
        self._ck(self.lib.nasr_set_learning_rate(self.h, float(lr)))

    # ------------------------------------------------------------------ host-buffer entry points
    @staticmethod
    def _batch(feats, seq_len, labels=None, label_len=None):
        feats = _f32(feats)
        assert feats.ndim == 3, 'features must be [B,T,F]'
        B, T, _ = feats.shape
        seq = _i32(np.asarray([int(x) for x in seq_len])).ravel()
        assert seq.size == B
        if labels is None:
            return feats, seq, None, None, B, T, 0
        labels = _i32(labels)
        if labels.ndim == 1:
            labels = labels.reshape(B, -1)
        ll = _i32(np.asarray([int(x) for x in label_len])).ravel()
        assert labels.shape[0] == B and ll.size == B
        return feats, seq, labels, ll, B, T, labels.shape[1]

    def logit_frames(self, T):
        return int(self.lib.nasr_logit_frames(self.h, int(T)))

    def forward(self, feats, seq_len):
        feats, seq, _, _, B, T, _ = self._batch(feats, seq_len)
        if feats.shape[2] != self.cfg.feature_size:
            raise ValueError(f'feature size {feats.shape[2]} != configured {self.cfg.feature_size}')
        out = np.empty((self.logit_frames(T), B, self.num_classes), np.float32)
        self._ck(self.lib.nasr_forward(self.h, _fp(feats), _ip(seq), B, T, _fp(out)))
        return out

    def loss(self, feats, seq_len, labels, label_len):
        feats, seq, labels, ll, B, T, Lmax = self._batch(feats, seq_len, labels, label_len)
        loss = c_float()
        nll = np.empty(B, np.float32)
        self._ck(self.lib.nasr_loss(self.h, _fp(feats), _ip(seq), _ip(labels), _ip(ll), B, T, Lmax, byref(loss),
                                    _fp(nll)))
        return float(loss.value), nll

    def loss_and_grads(self, feats, seq_len, labels, label_len):
        feats, seq, labels, ll, B, T, Lmax = self._batch(feats, seq_len, labels, label_len)
        loss = c_float()
        nll = np.empty(B, np.float32)
        grads = np.empty(self.param_count, np.float32)
        self._ck(self.lib.nasr_loss_and_grads(self.h, _fp(feats), _ip(seq), _ip(labels), _ip(ll), B, T, Lmax,
                                              byref(loss), _fp(nll), _fp(grads)))
        return float(loss.value), nll, grads

    def train_step(self, feats, seq_len, labels, label_len):
        feats, seq, labels, ll, B, T, Lmax = self._batch(feats, seq_len, labels, label_len)
        loss = c_float()
        self._ck(self.lib.nasr_train_step(self.h, _fp(feats), _ip(seq), _ip(labels), _ip(ll), B, T, Lmax, byref(loss)))
        return float(loss.value)

    def greedy_decode(self, feats, seq_len):
        feats, seq, _, _, B, T, _ = self._batch(feats, seq_len)
        Tp = self.logit_frames(T)
        ids = np.zeros((B, Tp), np.int32)
        lens = np.zeros(B, np.int32)
        self._ck(self.lib.nasr_greedy_decode(self.h, _fp(feats), _ip(seq), B, T, _ip(ids), _ip(lens)))
        return [ids[b, :lens[b]].tolist() for b in range(B)]

    # ------------------------------------------------------------------ resident-batch / data-parallel pieces
    def upload_batch(self, feats, seq_len, labels, label_len):
        feats, seq, labels, ll, B, T, Lmax = self._batch(feats, seq_len, labels, label_len)
        self._ck(self.lib.nasr_upload_batch(self.h, _fp(feats), _ip(seq), _ip(labels), _ip(ll), B, T, Lmax))

    @staticmethod
    def context_structure_ok(feats, seq, numcontext, numcep):
        """True when feats [B,T,(2*numcontext+1)*numcep] is what include_context (utils.py:8-21) + one constant pad value
        per utterance produce, so that the centre slice + the pad value determine it.  The first / last numcontext frames
        of EVERY utterance are checked exactly: that is where an array that went through rand_shift's roll-and-crop
        (dataset.py:23-31) differs (real neighbours instead of the pad value, and a pad read from a real sample); the
        interior is spot-checked."""
        B = feats.shape[0]
        w = 2 * numcontext + 1
        if numcontext < 1 or feats.shape[2] != w * numcep:
            return False
        pad = feats[:, 0, 0]
        c0 = numcontext * numcep
        rs = np.random.RandomState(0)
        for b in range(B):
            n = int(seq[b])
            edge = sorted(set(range(min(numcontext, n))) | set(range(max(0, n - numcontext), n)))
            t = np.asarray(edge + [rs.randint(n) for _ in range(4)], dtype=np.int64)
            src = t[:, None] + np.arange(w) - numcontext                      # [E, w]: source frame of each slot
            inside = (src >= 0) & (src < n)
            want = np.where(inside[:, :, None], feats[b, np.clip(src, 0, n - 1), c0:c0 + numcep], pad[b])
            if not np.array_equal(feats[b, t].reshape(len(t), w, numcep), want):
                return False
        return True

    def upload_batch_context(self, feats, seq_len, labels, label_len, numcontext, numcep):
        """Upload context-stacked features [B,T,(2*numcontext+1)*numcep] as their centre slice and rebuild the
        stacking on the device (include_context, utils.py:8-21).  Returns False (nothing uploaded) when the
        array does not have that structure (e.g. rand_shift cropped it), so the caller can upload it whole."""
        feats, seq, labels, ll, B, T, Lmax = self._batch(feats, seq_len, labels, label_len)
        if not self.context_structure_ok(feats, seq, numcontext, numcep):
            return False
        pad = np.ascontiguousarray(feats[:, 0, 0])
        centre = np.ascontiguousarray(feats[:, :, numcontext * numcep:(numcontext + 1) * numcep])
        self._ck(self.lib.nasr_upload_batch_context(self.h, _fp(centre), _fp(pad), int(numcontext), int(numcep),
                                                    _ip(seq), _ip(labels), _ip(ll), B, T, Lmax))
        return True

    def stage_batch(self, feats, seq_len, labels, label_len, numcontext=0, numcep=0):
        """Copy the NEXT batch towards the GPU while the current step runs (pinned staging + the handle's copy stream,
        include/nasr.h nasr_stage_batch); may be called from a loader thread.  With numcontext > 0 and features that have
        include_context's structure only the centre slice crosses PCIe.  Returns a ticket for commit_batch(), or None
        when no staging slot is free (upload the batch the synchronous way then)."""
        from ctypes import c_int
        feats, seq, labels, ll, B, T, Lmax = self._batch(feats, seq_len, labels, label_len)
        ticket = c_int(-1)
        if numcontext > 0 and self.context_structure_ok(feats, seq, numcontext, numcep):
            pad = np.ascontiguousarray(feats[:, 0, 0])
            centre = np.ascontiguousarray(feats[:, :, numcontext * numcep:(numcontext + 1) * numcep])
            rc = self.lib.nasr_stage_batch_context(self.h, _fp(centre), _fp(pad), int(numcontext), int(numcep), _ip(seq),
                                                   _ip(labels), _ip(ll), B, T, Lmax, byref(ticket))
        else:
            rc = self.lib.nasr_stage_batch(self.h, _fp(feats), _ip(seq), _ip(labels), _ip(ll), B, T, Lmax, byref(ticket))
        if rc == _lib.NASR_ERR_STATE:
            return None
        self._ck(rc)
        return int(ticket.value)

    def commit_batch(self, ticket):
        self._ck(self.lib.nasr_commit_batch(self.h, int(ticket)))

    def discard_batch(self, ticket):
        self._ck(self.lib.nasr_discard_batch(self.h, int(ticket)))

    def compute_grads(self):
        self._ck(self.lib.nasr_compute_grads(self.h))

    def apply_adam(self, grad_scale=1.0):
        self._ck(self.lib.nasr_apply_adam(self.h, float(grad_scale)))

    def get_grads(self):
        g = np.empty(self.param_count, np.float32)
        self._ck(self.lib.nasr_get_grads(self.h, _fp(g), g.size))
        return g

    def set_grads(self, flat):
        flat = _f32(flat).ravel()
        self._ck(self.lib.nasr_set_grads(self.h, _fp(flat), flat.size))

    def label_error_rate(self, hyps, labels, label_len):
        """mean over the batch of edit_distance(hyp, truth)/len(truth) (networks/tfnetwork.py:66-70)."""
        B = len(hyps)
        stride = max(1, max((len(h) for h in hyps), default=1))
        ids = np.zeros((B, stride), np.int32)
        lens = np.zeros(B, np.int32)
        for b, hy in enumerate(hyps):
            lens[b] = len(hy)
            ids[b, :len(hy)] = hy
        labels = _i32(labels).reshape(B, -1)
        ll = _i32(np.asarray([int(x) for x in label_len]))
        out = c_float()
        rc = self.lib.nasr_label_error_rate(_ip(ids), _ip(lens), stride, _ip(labels), _ip(ll), labels.shape[1], B,
                                            byref(out))
        if rc != 0:
            raise _lib.NasrError(rc, 'nasr_label_error_rate: bad arguments')
        return float(out.value)

    def set_step_decode(self, on, logits=False, greedy=True):
        """Per step: loss + greedy decode copied out behind the CTC kernels (on), with logits=True the logits themselves too,
        with greedy=False (and logits) those without the greedy decode (include/nasr.h: nasr_set_step_decode)."""
        mode = 0 if not on else ((1 if greedy or not logits else 0) | (2 if logits else 0))
        self._ck(self.lib.nasr_set_step_decode(self.h, mode))

    def step_logits(self, B, T):
        """[T',B,C] logits of the step just enqueued, as soon as its forward pass + CTC are done (the backward pass runs on)."""
        out = np.empty((self.logit_frames(T), B, self.num_classes), np.float32)
        self._ck(self.lib.nasr_get_step_logits(self.h, _fp(out)))
        return out

    def get_decoded(self, B, T):
        Tp = self.logit_frames(T)
        ids = np.zeros((B, Tp), np.int32)
        lens = np.zeros(B, np.int32)
        self._ck(self.lib.nasr_get_decoded(self.h, _ip(ids), _ip(lens)))
        return [ids[b, :lens[b]].tolist() for b in range(B)]

    def beam_search(self, logits_tm, seq_len, beam_width=100, merge_repeated=True):
        """tf.nn.ctc_beam_search_decoder defaults (networks/tfnetwork.py:61-64) on host logits [T',B,C].
        Returns (list of id lists, log-probabilities [B])."""
        lg = _f32(logits_tm)
        Tp, B, C = lg.shape
        seq = _i32(np.asarray([int(x) for x in seq_len]))
        ids = np.zeros((B, Tp), np.int32)
        lens = np.zeros(B, np.int32)
        logp = np.zeros(B, np.float32)
        rc = self.lib.nasr_ctc_beam_search(_fp(lg), _ip(seq), B, Tp, C, int(beam_width), int(bool(merge_repeated)),
                                           _ip(ids), _ip(lens), _fp(logp))
        if rc != 0:
            raise _lib.NasrError(rc, 'nasr_ctc_beam_search: bad arguments')
        return [ids[b, :lens[b]].tolist() for b in range(B)], logp

    def get_loss(self):
        loss = c_float()
        self._ck(self.lib.nasr_get_loss(self.h, byref(loss)))
        return float(loss.value)

    def step_void(self):
        """True when the step just applied was void on every rank (some rank's persistent recurrence aborted; Adam was
        a no-op everywhere): run it again.  Synchronises."""
        from ctypes import c_int
        v = c_int()
        self._ck(self.lib.nasr_step_void(self.h, byref(v)))
        return bool(v.value)

    def step_results(self, B, T):
        """(loss, forward_fault, hypotheses) of the step just enqueued, as soon as its forward pass + CTC are done - the
        backward pass, the exchange and Adam keep running (include/nasr.h, nasr_get_step_results)."""
        from ctypes import c_int
        Tp = self.logit_frames(T)
        ids = np.zeros((B, Tp), np.int32)
        lens = np.zeros(B, np.int32)
        loss, fault = c_float(), c_int()
        self._ck(self.lib.nasr_get_step_results(self.h, byref(loss), byref(fault), _ip(ids), _ip(lens)))
        return float(loss.value), bool(fault.value), [ids[b, :lens[b]].tolist() for b in range(B)]

    def settle_step(self, previous=False):
        """True when the latest (or, previous=True, the one-before-latest) optimiser step was void on every rank."""
        from ctypes import c_int
        v = c_int()
        self._ck(self.lib.nasr_settle_step(self.h, int(bool(previous)), byref(v)))
        return bool(v.value)

    @property
    def wgrad_overlap(self):
        """True when the upper layers' weight gradients run beside the persistent BPTT launch of the layer below."""
        return bool(self.lib.nasr_get_wgrad_overlap(self.h))

    def set_wgrad_overlap(self, on):
        self._ck(self.lib.nasr_set_wgrad_overlap(self.h, int(bool(on))))

    def diag_bucket_traffic(self, i, stream, nblocks, passes):
        """Diagnostics: a ring-all-reduce-shaped kernel over bucket i on `stream`, behind the bucket's event (include/nasr.h)."""
        from ctypes import c_void_p
        self._ck(self.lib.nasr_diag_bucket_traffic(self.h, int(i), c_void_p(int(stream)), int(nblocks), int(passes)))

    def step_token(self):
        """Sequence number of the optimiser step apply_adam() enqueued last (include/nasr.h, nasr_step_token)."""
        return int(self.lib.nasr_step_token(self.h))

    def settle_token(self, token):
        """True when optimiser step `token` was void on every rank; waits for the end of exactly that step."""
        from ctypes import c_int
        v = c_int()
        self._ck(self.lib.nasr_settle_token(self.h, int(token), byref(v)))
        return bool(v.value)

    def resident_frames(self):
        n = c_int64()
        self._ck(self.lib.nasr_resident_frames(self.h, byref(n)))
        return n.value

    def resident_rows(self):
        """Rows the operand passes and GEMMs cover for the resident batch: sum(seq_len) when the ragged batch was compacted
        (nasr.h: nasr_set_row_compaction), T x padded B otherwise."""
        n = c_int64()
        self._ck(self.lib.nasr_resident_rows(self.h, byref(n)))
        return n.value

    def set_row_compaction(self, on):
        self._ck(self.lib.nasr_set_row_compaction(self.h, int(bool(on))))

    def grad_device_ptr(self):
        return int(self.lib.nasr_grad_device_ptr(self.h)), int(self.lib.nasr_grad_device_count(self.h))

    def grad_tensor(self):
        """The flat device gradient buffer as a torch tensor ALIAS (no copy), for torch.distributed
        all-reduce over RCCL.  torch is plumbing here: it only wraps the pointer."""
        import torch
        ptr, n = self.grad_device_ptr()

        class _Ext:
            __cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (ptr, False), 'version': 3,
                                        'strides': None}
        return torch.as_tensor(_Ext(), device='cuda')

    def grad_buckets(self):
        """[(offset, count)] in floats from grad_device_ptr(): contiguous pieces of the gradient buffer in the order
        compute_grads() completes them (include/nasr.h, "Overlapping the exchange")."""
        out = []
        n = self.lib.nasr_grad_bucket_count(self.h)
        if n < 0:
            self._ck(n)
        for i in range(n):
            o, c = c_int64(), c_int64()
            self._ck(self.lib.nasr_grad_bucket(self.h, i, byref(o), byref(c)))
            out.append((int(o.value), int(c.value)))
        return out

    # ------------------------------------------------------------------ in-library RCCL exchange (include/nasr.h nasr_comm_*)
    def comm_unique_id(self):
        """128 bytes from ncclGetUniqueId (rank 0 calls this and hands them to the other ranks)."""
        buf = ctypes.create_string_buffer(128)
        rc = self.lib.nasr_comm_unique_id(buf)
        if rc != 0:
            msg = self.lib.nasr_last_error(None)
            raise _lib.NasrError(rc, msg.decode() if msg else 'nasr_comm_unique_id failed')
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        assert len(unique_id) == 128
        self._ck(self.lib.nasr_comm_init(self.h, ctypes.c_char_p(bytes(unique_id)), int(rank), int(nranks)))

    def comm_size(self):
        return int(self.lib.nasr_comm_size(self.h))

    def comm_allreduce_grads(self):
        self._ck(self.lib.nasr_comm_allreduce_grads(self.h))

    def comm_mean(self, values):
        v = np.ascontiguousarray(values, np.float32).copy()
        self._ck(self.lib.nasr_comm_mean(self.h, _fp(v), v.size))
        return v.tolist()

    def comm_destroy(self):
        self._ck(self.lib.nasr_comm_destroy(self.h))

    def set_bucket_defer(self, on):
        """Hold each gradient bucket's event back over the next persistent BPTT launch (include/nasr.h)."""
        self._ck(self.lib.nasr_set_bucket_defer(self.h, int(bool(on))))

    def bucket_wait(self, i, stream):
        """Makes HIP stream `stream` (raw handle) wait until the compute_grads() issued before has completed bucket i."""
        self._ck(self.lib.nasr_grad_bucket_wait(self.h, int(i), c_void_p(int(stream))))

    # ------------------------------------------------------------------ measurement
    def set_profiling(self, on):
        self._ck(self.lib.nasr_set_profiling(self.h, int(bool(on))))

    def set_graph_mode(self, on):
        self._ck(self.lib.nasr_set_graph_mode(self.h, int(bool(on))))

    def set_dropout_state(self, seed, counter):
        """Pins the keep-masks of the dense stages: forward pass number `counter` of stream `seed` (every forward pass
        uses the current counter, then increments it)."""
        self._ck(self.lib.nasr_set_dropout_state(self.h, int(seed) & 0xFFFFFFFF, int(counter) & 0xFFFFFFFF))

    def dropout_state(self):
        s, c = c_uint32(), c_uint32()
        self._ck(self.lib.nasr_get_dropout_state(self.h, byref(s), byref(c)))
        return int(s.value), int(c.value)

    @property
    def recurrence_mode(self):
        """'persistent' (one launch per layer pass, lstm_persist.hip), 'wide-persistent' (Hp = 2048: one persistent launch per
        direction and pass, lstm_wide.hip) or 'per-step' (lstm.hip)."""
        return ('per-step', 'persistent', 'wide-persistent')[int(self.lib.nasr_get_recurrence_mode(self.h))]

    def persist_stats(self):
        """(aborts, re-arms) of the persistent recurrence on this handle (include/nasr.h, nasr_get_persist_stats)."""
        from ctypes import c_int
        a, r = c_int(), c_int()
        self._ck(self.lib.nasr_get_persist_stats(self.h, byref(a), byref(r)))
        return int(a.value), int(r.value)

    def set_recurrence_mode(self, persistent):
        self._ck(self.lib.nasr_set_recurrence_mode(self.h, int(bool(persistent))))

    def phase_times(self):
        pt = _lib.PhaseTimes()
        self._ck(self.lib.nasr_get_phase_times(self.h, byref(pt)))
        return pt.as_dict()
