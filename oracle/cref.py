"""ctypes view of the C/OpenMP restatement (oracle/cref/nasr_cref.c).  Test infrastructure only: imported by tests/
and by bench.py's cpu_baseline leg."""
import ctypes
import os

import numpy as np

from . import build_cref

MERGE = {'none': 0, 'stack_reshape': 1, 'concat': 2}


class Spec(ctypes.Structure):
    _fields_ = [('feature_size', ctypes.c_int32), ('hidden', ctypes.c_int32), ('num_layers', ctypes.c_int32),
                ('bidirectional', ctypes.c_int32), ('merge', ctypes.c_int32), ('num_classes', ctypes.c_int32),
                ('forget_bias', ctypes.c_float),
                # DeepSpeech family (networks/deepspeech.py): dense stages around the stack, hash-defined dropout
                ('num_pre', ctypes.c_int32), ('pre_width', ctypes.c_int32 * 3), ('post_width', ctypes.c_int32),
                ('relu_clip', ctypes.c_float), ('dropout', ctypes.c_float * 4), ('drop_seed', ctypes.c_uint32),
                ('drop_counter', ctypes.c_uint32), ('use_dropout', ctypes.c_int32)]


def c_spec(spec, drop=None):
    cs = Spec(spec.feature_size, spec.hidden, spec.num_layers, int(spec.bidirectional), MERGE[spec.merge],
              spec.num_classes, float(spec.forget_bias))
    pre = tuple(getattr(spec, 'pre', ()) or ())
    cs.num_pre = len(pre)
    for i, w in enumerate(pre):
        cs.pre_width[i] = int(w)
    cs.post_width = int(getattr(spec, 'post', 0) or 0)
    cs.relu_clip = float(getattr(spec, 'relu_clip', 20.0))
    if drop is not None and (pre or cs.post_width):
        for i in range(4):
            cs.dropout[i] = float(spec.drop_p(i))
        cs.drop_seed, cs.drop_counter, cs.use_dropout = int(drop[0]) & 0xFFFFFFFF, int(drop[1]) & 0xFFFFFFFF, 1
    return cs


_lib = None


def load():
    global _lib
    if _lib is None:
        path = build_cref.OUT if os.path.exists(build_cref.OUT) else build_cref.build()
        lib = ctypes.CDLL(path)
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
        lib.cref_param_count.restype = ctypes.c_int64
        lib.cref_param_count.argtypes = [ctypes.POINTER(Spec)]
        lib.cref_num_threads.restype = ctypes.c_int
        lib.cref_set_threads.argtypes = [ctypes.c_int]
        lib.cref_loss_and_grads.restype = ctypes.c_int
        lib.cref_loss_and_grads.argtypes = [ctypes.POINTER(Spec), fp, fp, ip, ip, ip, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, fp, fp, fp, fp]
        _lib = lib
    return _lib


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (a container with a
    16-CPU quota on a 256-thread host must not run 256 OpenMP threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def set_threads(n):
    load().cref_set_threads(int(n))


def num_threads():
    return int(load().cref_num_threads())


def loss_and_grads(spec, flat_params, feats, seq_len, labels, label_len, want_grads=True, want_logits=False, drop=None):
    """spec: oracle.nasr_oracle.ModelSpec; drop = (seed, counter) of the dropout masks of the DeepSpeech family, or None
    for no dropout.  Returns (loss, nll [B], grads flat | None, logits [T',B,C] | None)."""
    lib = load()
    cs = c_spec(spec, drop)
    P = np.ascontiguousarray(flat_params, np.float32)
    assert P.size == lib.cref_param_count(ctypes.byref(cs))
    X = np.ascontiguousarray(feats, np.float32)
    B, T, _ = X.shape
    sl = np.ascontiguousarray(seq_len, np.int32)
    lab = np.ascontiguousarray(labels, np.int32).reshape(B, -1)
    ll = np.ascontiguousarray(label_len, np.int32)
    loss = ctypes.c_float()
    nll = np.zeros(B, np.float32)
    g = np.zeros(P.size, np.float32) if want_grads else None
    Tp = spec.logit_frames(T)
    lg = np.zeros((Tp, B, spec.num_classes), np.float32) if want_logits else None
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    rc = lib.cref_loss_and_grads(ctypes.byref(cs), P.ctypes.data_as(fp), X.ctypes.data_as(fp), sl.ctypes.data_as(ip),
                                 lab.ctypes.data_as(ip), ll.ctypes.data_as(ip), B, T, lab.shape[1], ctypes.byref(loss),
                                 nll.ctypes.data_as(fp), g.ctypes.data_as(fp) if want_grads else None,
                                 lg.ctypes.data_as(fp) if want_logits else None)
    if rc == -3:
        raise ValueError('Not enough time for target transition sequence')
    if rc != 0:
        raise RuntimeError('cref_loss_and_grads failed: %d' % rc)
    return float(loss.value), nll, g, lg
