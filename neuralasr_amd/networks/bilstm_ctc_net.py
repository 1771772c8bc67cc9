"""BiLstmCTCNet (reference: networks/bilstm_ctc_net.py:6-52): ONE bidirectional BasicLSTMCell(500,
forget_bias=1.0) layer; the (fw, bw) output tuple goes through tf.reshape(outputs, [-1, 500]), i.e. the
literal stack-reshape index map, W [500, C], b [C], time-major logits [2T, B, C].
`network=networks.bilstm_ctc_net.BiLstmCTCNet` in a NeuralASR config selects this class."""
from .hipnetwork import HipNetwork


class BiLstmCTCNet(HipNetwork):
    num_hidden = 500
    num_layers = 1
    bidirectional = True
    merge = 'stack_reshape'


class BiLstmConcatCTCNet(HipNetwork):
    """1x500 bidirectional with the outputs concatenated (tf.concat(outputs, 2), as networks/deepspeech.py:103
    does): the self-consistent variant of the net above.  No reference counterpart."""
    num_hidden = 500
    num_layers = 1
    bidirectional = True
    merge = 'concat'


class BiLstm3x500CTCNet(HipNetwork):
    """3x500 bidirectional stack, concat merge: the benchmark shape BASELINE.json configs[1] names.
    No reference counterpart."""
    num_hidden = 500
    num_layers = 3
    bidirectional = True
    merge = 'concat'
