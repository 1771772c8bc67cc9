"""Training loop (reference: train.py:14-63): epochs x batches, cumulative wall time, checkpoint + log +
one validation batch every report_step, same log-line formats.

  python -m neuralasr_amd.train <config>
  python -m torch.distributed.run --nproc-per-node N -m neuralasr_amd.train <config>   # num_gpus = N"""
import argparse
import os
import time

from .config import Config
from .dataset import DataSet
from .logger import get_logger

logger = get_logger()


def train_model(dataTrain, datavalid, config, prefetch=2):
    """The loop of train.py:14-47.  With a network that offers begin_step / finish_step / stage_batch (HipNetwork) the
    next batch is loaded, padded and staged towards the GPU BETWEEN the two halves of a step, i.e. while the device runs
    that step: one host thread, no idle device.  `prefetch` = 0 keeps the reference's order (load, then train) with
    synchronous uploads; other networks get a loader thread that runs `prefetch` batches ahead."""
    logger.info('Batch Dimensions: ' + str(dataTrain.get_feature_shape()))
    logger.info('Label Dimensions: ' + str(dataTrain.get_label_shape()))
    network = config.load_network(fortraining=True)
    spent, loss_sum, ler_sum = 0.0, 0.0, 0.0     # train_time_sec is never reset (train.py:20,26,34)
    counted = 0                                  # steps of the window whose values count (all of them, normally)
    split = prefetch and all(hasattr(network, a) for a in ('begin_step', 'finish_step', 'stage_batch'))
    lazy_ok = split and hasattr(network, 'mean_over_ranks')      # finish_step(lazy=True) is HipNetwork's

    window = []                                   # (loss, mean_ler or a handle with .result()) of the steps since the last log line

    def report(loss, mean_ler):
        nonlocal loss_sum, ler_sum, counted
        window.append((loss, mean_ler))
        if network.global_step % config.report_step == 0:
            lazy = False
            for lo, le in window:
                if hasattr(le, 'result'):             # HipNetwork: the step's beam search ran on host threads meanwhile
                    le, lazy = le.result(), True
                # a step whose forward pass was void on some rank (HipNetwork: a persistent-recurrence abort, repeated one
                # call later) reports NaN on every rank: it stays out of the window's means instead of turning them into NaN
                if lo == lo and le == le:
                    loss_sum += lo
                    ler_sum += le
                    counted += 1
            del window[:]
            if lazy and hasattr(network, 'mean_over_ranks'):
                ler_sum = network.mean_over_ranks(ler_sum)     # lazily decoded LERs are local until here (one collective per log line)
            network.save_checkpoint()
            # train.py:32-34 divides by report_step; `counted` equals it unless a void step was left out
            den = counted if 0 < counted < config.report_step else config.report_step
            logger.info('Step: %04d' % network.global_step + ', cost = %.4f' % (loss_sum / den) +
                        ', ler = %.4f' % (ler_sum / den) + ', time = %.4f' % spent)
            loss_sum = ler_sum = 0.0
            counted = 0
            if datavalid:
                if not datavalid.has_more_batches():
                    datavalid.reset_epoch()
                vm, vl, vs, vll = datavalid.get_next_batch()
                vloss, vler = network.validate(vm, vl, vs, vll)
                logger.info('Valid: cost = %.4f' % vloss + ', ler = %.4f' % vler)

    if split:
        def all_batches():                                # epochs x batches, in the reference's order
            for _ in range(config.epochs):
                while dataTrain.has_more_batches():
                    yield dataTrain.get_next_batch()
                dataTrain.reset_epoch()
        it = all_batches()
        t0 = time.time()
        cur = next(it, None)
        while cur is not None:
            network.begin_step(*cur)
            nxt = next(it, None)                          # loaded, padded and staged under the step the device is running
            if nxt is not None:
                network.stage_batch(*nxt)
            loss, mean_ler = network.finish_step(lazy=True) if lazy_ok else network.finish_step()
            spent += time.time() - t0
            report(loss, mean_ler)
            t0 = time.time()
            cur = nxt
        network.discard_staged()
        logger.info('Finished training!!!')
        return network
    for _ in range(config.epochs):
        batches = dataTrain.prefetch(prefetch) if prefetch else iter(dataTrain.get_next_batch, None)
        while True:
            t0 = time.time()
            if not prefetch and not dataTrain.has_more_batches():
                break
            try:
                mfccs, labels, seq_len, labels_len = next(batches)
            except StopIteration:
                break
            loss, mean_ler = network.train(mfccs, labels, seq_len, labels_len)
            spent += time.time() - t0
            report(loss, mean_ler)
        dataTrain.reset_epoch()
        if hasattr(network, 'discard_staged'):
            network.discard_staged()
    logger.info('Finished training!!!')
    return network


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train speech recognizer on featurized mfcc files.')
    ap.add_argument('config', help='Configuration file.')
    args = ap.parse_args(argv)
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    config = Config(args.config, isTraining=True)
    dataTrain = DataSet(config.train_input, config)
    dataValid = None
    if config.test_input:
        config_test = Config(args.config, isTraining=True)
        config_test.epochs = None
        dataValid = DataSet(config_test.test_input, config_test)
    train_model(dataTrain, dataValid, config)


if __name__ == '__main__':
    main()
