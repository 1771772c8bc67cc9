"""Pure-NumPy helpers of the reference's utils.py that sit next to the hot path (the audio front end —
librosa / python_speech_features — is out of scope)."""
import numpy as np


def include_context(audio_mfcc, numcontext, numcep):
    """Stack numcontext frames either side of every frame (reference: utils.py:8-21): [T,numcep] ->
    [T,(2*numcontext+1)*numcep], zero frames beyond the ends."""
    audio_mfcc = np.asarray(audio_mfcc)
    T = audio_mfcc.shape[0]
    pad = np.zeros((numcontext, numcep), dtype=audio_mfcc.dtype)
    ext = np.concatenate((pad, audio_mfcc, pad))
    win = 2 * numcontext + 1
    out = np.empty((T, win * numcep), dtype=audio_mfcc.dtype)
    for w in range(win):
        out[:, w * numcep:(w + 1) * numcep] = ext[w:w + T]
    return out


def sparse_tuple_from(sequences, output_lengths):
    """dense padded labels + lengths -> (indices int64 [n,2], values int32 [n], shape int64 [2])
    (reference: utils.py:44-58), the feed of tf.sparse_placeholder.  The HIP path takes the dense form
    directly; this is kept for callers that expect the tuple."""
    idx, vals = [], []
    for n, seq in enumerate(sequences):
        L = int(output_lengths[n])
        idx.extend((n, k) for k in range(L))
        vals.extend(seq[:L])
    indices = np.asarray(idx, dtype=np.int64).reshape(-1, 2)
    values = np.asarray(vals, dtype=np.int32)
    shape = np.asarray([len(sequences), indices[:, 1].max() + 1], dtype=np.int64)
    return indices, values, shape
