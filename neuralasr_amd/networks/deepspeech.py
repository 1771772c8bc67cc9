"""DeepSpeech (reference: networks/deepspeech.py:6-132): three clipped-ReLU dense layers with dropout (2048, 2048, 4096),
one bidirectional BasicLSTMCell(2048, forget_bias=1.0) layer whose directions are concatenated, a fourth clipped-ReLU
dense layer (2048) and the affine output layer; time-major logits [T, B, C].  The locals of the reference's
create_network are class attributes here.  `network=networks.deepspeech.DeepSpeech` selects it.

tf.nn.dropout is part of the graph whether or not it trains (deepspeech.py:50,59,68,113 do not look at is_training) and
the DropoutWrappers around the cells have keep probability 1 (dropout[3] = dropout[4] = 0): both are kept as they are.
TensorFlow's random stream is not reproducible; the keep-masks are a hash of (seed 4567, forward-pass counter, layer,
frame, utterance, unit), see neuralasr_amd/csrc/dense.hip."""
import numpy as np

from ..engine import Engine
from .hipnetwork import HipNetwork


class DeepSpeech(HipNetwork):
    n_hidden = 2048                       # n_hidden_1 = n_hidden_2 = n_hidden_5
    n_cell_dim = 2048
    relu_clip = 20.0
    stddev = 0.046875
    random_seed = 4567
    dropout = (0.05, 0.05, 0.05, 0.05)    # layers 1, 2, 3 and 5 (the reference's list also holds the two 0.0 of the cells)
    num_layers = 1
    bidirectional = True
    merge = 'concat'

    @classmethod
    def pre_widths(cls):
        return (cls.n_hidden, cls.n_hidden, 2 * cls.n_cell_dim)     # n_hidden_3 = 2 * n_cell_dim

    @property
    def num_hidden(self):
        return self.n_cell_dim

    def make_engine(self, config, device, stream):
        e = Engine(config.feature_size, self.n_cell_dim, self.num_layers, True, 'concat', self.num_classes,
                   learning_rate=config.learningrate, device_id=device, stream=stream, pre=self.pre_widths(),
                   post=self.n_hidden, relu_clip=self.relu_clip, dropout=self.dropout)
        e.set_dropout_state(self.random_seed, 0)
        return e

    def initial_params(self, tensors, seed):
        """deepspeech.py:43-121: b_i ~ N(0, stddev), h1 / h6 xavier-normal, h2 / h3 / h5 ~ N(0, stddev); the cells keep
        TF's defaults (glorot-uniform kernels, zero biases)."""
        rs = np.random.RandomState(seed)
        chunks = []
        for name, _, rows, cols in tensors:
            if name.endswith('kernel'):
                lim = np.sqrt(6.0 / (rows + cols))
                chunks.append(rs.uniform(-lim, lim, size=rows * cols))
            elif name in ('h1', 'h6'):
                chunks.append(rs.randn(rows * cols) * np.sqrt(2.0 / (rows + cols)))
            elif name[0] in 'hb' and name[1:].isdigit():
                chunks.append(rs.randn(rows * cols) * self.stddev)
            else:
                chunks.append(np.zeros(rows * cols))
        return np.concatenate(chunks).astype(np.float32)
