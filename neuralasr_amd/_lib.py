"""ctypes binding of libnasr.so (include/nasr.h).  There is no CPU fallback: if the HIP library is
missing or no gfx950 device is usable, everything here raises."""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libnasr.so')

NASR_OK = 0
NASR_ERR_ARG = -1
NASR_ERR_HIP = -2
NASR_ERR_INFEASIBLE = -3
NASR_ERR_STATE = -4

MERGE_NONE, MERGE_STACK_RESHAPE, MERGE_CONCAT = 0, 1, 2
MERGE_BY_NAME = {'none': MERGE_NONE, 'stack_reshape': MERGE_STACK_RESHAPE, 'concat': MERGE_CONCAT}

# every symbol include/nasr.h declares (tests/test_abi.py checks the .so exports exactly these)
SYMBOLS = [
    'nasr_create', 'nasr_destroy', 'nasr_last_error', 'nasr_backend', 'nasr_synchronize', 'nasr_param_count',
    'nasr_num_tensors', 'nasr_tensor_info', 'nasr_set_params', 'nasr_get_params', 'nasr_set_adam_state',
    'nasr_get_adam_state', 'nasr_set_learning_rate', 'nasr_train_step', 'nasr_forward', 'nasr_logit_frames',
    'nasr_loss', 'nasr_loss_and_grads', 'nasr_greedy_decode', 'nasr_upload_batch', 'nasr_compute_grads',
    'nasr_grad_device_ptr', 'nasr_grad_device_count', 'nasr_grad_bucket_count', 'nasr_grad_bucket',
    'nasr_grad_bucket_wait', 'nasr_apply_adam', 'nasr_get_grads', 'nasr_set_grads',
    'nasr_upload_batch_context', 'nasr_label_error_rate', 'nasr_set_step_decode', 'nasr_get_decoded', 'nasr_ctc_beam_search', 'nasr_get_loss', 'nasr_resident_frames',
    'nasr_set_profiling', 'nasr_get_phase_times', 'nasr_set_graph_mode',
    'nasr_get_recurrence_mode', 'nasr_set_recurrence_mode', 'nasr_set_dropout_state', 'nasr_get_dropout_state',
    'nasr_step_void', 'nasr_get_persist_stats', 'nasr_stage_batch', 'nasr_stage_batch_context', 'nasr_commit_batch',
    'nasr_discard_batch', 'nasr_set_bucket_defer', 'nasr_comm_unique_id', 'nasr_comm_init', 'nasr_comm_size',
    'nasr_comm_allreduce_grads', 'nasr_comm_mean', 'nasr_comm_destroy', 'nasr_get_step_results', 'nasr_settle_step',
    'nasr_step_token', 'nasr_settle_token', 'nasr_diag_bucket_traffic', 'nasr_get_step_logits',
    'nasr_set_wgrad_overlap', 'nasr_get_wgrad_overlap', 'nasr_set_row_compaction', 'nasr_resident_rows',
]


class ModelCfg(Structure):
    _fields_ = [('feature_size', c_int32), ('hidden', c_int32), ('num_layers', c_int32), ('bidirectional', c_int32),
                ('merge', c_int32), ('num_classes', c_int32), ('forget_bias', c_float), ('learning_rate', c_float),
                ('beta1', c_float), ('beta2', c_float), ('epsilon', c_float),
                ('num_pre', c_int32), ('pre_width', c_int32 * 3), ('post_width', c_int32), ('relu_clip', c_float),
                ('dropout', c_float * 4)]


class PhaseTimes(Structure):
    _fields_ = [('pack_ms', c_float), ('xproj_ms', c_float), ('rec_fwd_ms', c_float), ('proj_ctc_ms', c_float),
                ('proj_bwd_ms', c_float), ('rec_bwd_ms', c_float), ('wgrad_ms', c_float), ('adam_ms', c_float),
                ('total_ms', c_float), ('rec_fwd_launches', c_int32), ('rec_bwd_launches', c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class NasrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libnasr error {code}: {msg}')
        self.code = code


class InfeasibleLabelError(NasrError, ValueError):
    """CTC: "Not enough time for target transition sequence" (TF raises InvalidArgumentError)."""


_lib = None


def load():
    """dlopen libnasr.so, building it first if the sources are newer (dev container).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} is missing: run `python -m neuralasr_amd.build` (hipcc, gfx950). '
                          'neuralasr_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    fp, ip = POINTER(c_float), POINTER(c_int32)
    H = c_void_p
    sig = {
        'nasr_create': (c_int, [POINTER(ModelCfg), c_int, c_void_p, POINTER(H)]),
        'nasr_destroy': (c_int, [H]),
        'nasr_last_error': (c_char_p, [H]),
        'nasr_backend': (c_char_p, [H]),
        'nasr_synchronize': (c_int, [H]),
        'nasr_param_count': (c_int64, [H]),
        'nasr_num_tensors': (c_int, [H]),
        'nasr_tensor_info': (c_int, [H, c_int, POINTER(c_char * 64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
        'nasr_set_params': (c_int, [H, fp, c_int64]),
        'nasr_get_params': (c_int, [H, fp, c_int64]),
        'nasr_set_adam_state': (c_int, [H, fp, fp, c_int64, c_int64]),
        'nasr_get_adam_state': (c_int, [H, fp, fp, c_int64, POINTER(c_int64)]),
        'nasr_set_learning_rate': (c_int, [H, c_float]),
        'nasr_train_step': (c_int, [H, fp, ip, ip, ip, c_int, c_int, c_int, fp]),
        'nasr_forward': (c_int, [H, fp, ip, c_int, c_int, fp]),
        'nasr_logit_frames': (c_int, [H, c_int]),
        'nasr_loss': (c_int, [H, fp, ip, ip, ip, c_int, c_int, c_int, fp, fp]),
        'nasr_loss_and_grads': (c_int, [H, fp, ip, ip, ip, c_int, c_int, c_int, fp, fp, fp]),
        'nasr_greedy_decode': (c_int, [H, fp, ip, c_int, c_int, ip, ip]),
        'nasr_upload_batch': (c_int, [H, fp, ip, ip, ip, c_int, c_int, c_int]),
        'nasr_upload_batch_context': (c_int, [H, fp, fp, c_int, c_int, ip, ip, ip, c_int, c_int, c_int]),
        'nasr_compute_grads': (c_int, [H]),
        'nasr_grad_device_ptr': (c_void_p, [H]),
        'nasr_grad_device_count': (c_int64, [H]),
        'nasr_grad_bucket_count': (c_int, [H]),
        'nasr_grad_bucket': (c_int, [H, c_int, POINTER(c_int64), POINTER(c_int64)]),
        'nasr_grad_bucket_wait': (c_int, [H, c_int, c_void_p]),
        'nasr_apply_adam': (c_int, [H, c_float]),
        'nasr_get_grads': (c_int, [H, fp, c_int64]),
        'nasr_set_grads': (c_int, [H, fp, c_int64]),
        'nasr_label_error_rate': (c_int, [ip, ip, c_int, ip, ip, c_int, c_int, fp]),
        'nasr_set_step_decode': (c_int, [H, c_int]),
        'nasr_get_decoded': (c_int, [H, ip, ip]),
        'nasr_ctc_beam_search': (c_int, [fp, ip, c_int, c_int, c_int, c_int, c_int, ip, ip, fp]),
        'nasr_get_loss': (c_int, [H, fp]),
        'nasr_resident_frames': (c_int, [H, POINTER(c_int64)]),
        'nasr_resident_rows': (c_int, [H, POINTER(c_int64)]),
        'nasr_set_row_compaction': (c_int, [H, c_int]),
        'nasr_set_profiling': (c_int, [H, c_int]),
        'nasr_get_phase_times': (c_int, [H, POINTER(PhaseTimes)]),
        'nasr_set_graph_mode': (c_int, [H, c_int]),
        'nasr_get_recurrence_mode': (c_int, [H]),
        'nasr_set_recurrence_mode': (c_int, [H, c_int]),
        'nasr_set_dropout_state': (c_int, [H, c_uint32, c_uint32]),
        'nasr_get_dropout_state': (c_int, [H, POINTER(c_uint32), POINTER(c_uint32)]),
        'nasr_step_void': (c_int, [H, POINTER(c_int)]),
        'nasr_get_persist_stats': (c_int, [H, POINTER(c_int), POINTER(c_int)]),
        'nasr_stage_batch': (c_int, [H, fp, ip, ip, ip, c_int, c_int, c_int, POINTER(c_int)]),
        'nasr_stage_batch_context': (c_int, [H, fp, fp, c_int, c_int, ip, ip, ip, c_int, c_int, c_int, POINTER(c_int)]),
        'nasr_commit_batch': (c_int, [H, c_int]),
        'nasr_discard_batch': (c_int, [H, c_int]),
        'nasr_set_bucket_defer': (c_int, [H, c_int]),
        'nasr_comm_unique_id': (c_int, [c_void_p]),
        'nasr_comm_init': (c_int, [H, c_void_p, c_int, c_int]),
        'nasr_comm_size': (c_int, [H]),
        'nasr_comm_allreduce_grads': (c_int, [H]),
        'nasr_comm_mean': (c_int, [H, fp, c_int]),
        'nasr_comm_destroy': (c_int, [H]),
        'nasr_get_step_results': (c_int, [H, fp, POINTER(c_int), ip, ip]),
        'nasr_settle_step': (c_int, [H, c_int, POINTER(c_int)]),
        'nasr_step_token': (c_int64, [H]),
        'nasr_settle_token': (c_int, [H, c_int64, POINTER(c_int)]),
        'nasr_diag_bucket_traffic': (c_int, [H, c_int, c_void_p, c_int, c_int]),
        'nasr_get_step_logits': (c_int, [H, fp]),
        'nasr_set_wgrad_overlap': (c_int, [H, c_int]),
        'nasr_get_wgrad_overlap': (c_int, [H]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, handle, rc):
    if rc == NASR_OK:
        return
    msg = lib.nasr_last_error(handle)
    msg = msg.decode() if msg else ''
    if rc == NASR_ERR_INFEASIBLE:
        raise InfeasibleLabelError(rc, msg)
    raise NasrError(rc, msg)
