"""The pickled training record (reference: audiosample.py:3-12).  preprocess_mfcc.py pickles instances
of `audiosample.AudioSample`; dataset.load_pkl resolves that module path to this class, so the same
.pkl files load unchanged."""


class AudioSample(object):
    """id, mfcc float32 [T,F], labels int32 [L], transcription str."""

    def __init__(self, id, mfcc, labels, transcription):
        self.id = id
        self.mfcc = mfcc
        self.labels = labels
        self.transcription = transcription
