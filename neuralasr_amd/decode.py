"""Evaluation over the test list, one utterance at a time (reference: decode.py:14-54)."""
import argparse
import time

import numpy as np

from .config import Config
from .dataset import DataSet
from .logger import get_logger

logger = get_logger()


def decode(dataTest, config):
    logger.info('Batch Dimensions: ' + str(dataTest.get_feature_shape()))
    logger.info('Label Dimensions: ' + str(dataTest.get_label_shape()))
    network = config.load_network(fortraining=False)
    steps, spent, loss_sum, ler_sum = 0, 0.0, 0.0, 0.0
    while dataTest.has_more_batches():
        steps += 1
        t0 = time.time()
        mfccs, labels, seq_len, labels_len = dataTest.get_next_batch()
        output, loss, ler = network.evaluate(mfccs, labels, seq_len, labels_len)
        logger.info('Valid: batch_cost = %.4f' % loss + ', batch_ler = %.4f' % ler)
        spent += time.time() - t0
        loss_sum += loss
        ler_sum += ler
        logger.info('Decoded: ' + config.symbols.convert_to_str(np.asarray(output)))
        logger.info('Original: ' + config.symbols.convert_to_str(np.asarray(labels[0])))
    logger.info('Finished Decoding!!!')
    logger.info('Decoded Time = %.4fs, avg_loss = %.4f, avg_ler = %.4f' % (spent, loss_sum / steps, ler_sum / steps))


def main(argv=None):
    ap = argparse.ArgumentParser(description='Decode test data using trained model.')
    ap.add_argument('config', help='Configuration file.')
    args = ap.parse_args(argv)
    config = Config(args.config, True)
    config.batch_size = 1
    config.epochs = 1
    config.rand_shift = 0
    decode(DataSet(config.test_input, config), config)


if __name__ == '__main__':
    main()
