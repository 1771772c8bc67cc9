"""LstmCTCNet (reference: networks/lstm_ctc_net.py:6-47): 3x LSTMCell(500) in a MultiRNNCell under
dynamic_rnn (unidirectional), W [500, C], b [C], time-major logits [T, B, C]."""
from .hipnetwork import HipNetwork


class LstmCTCNet(HipNetwork):
    num_hidden = 500
    num_layers = 3
    bidirectional = False
    merge = 'none'


class SmallLstmCTCNet(HipNetwork):
    """1x128 unidirectional: the plumbing shape BASELINE.json configs[0] names."""
    num_hidden = 128
    num_layers = 1
    bidirectional = False
    merge = 'none'
