"""Run configuration (reference: config.py:11-113): the same INI files (ExtendedInterpolation,
sections Parameters / Train / Test / MFCC Featurizer), the same derived fields
(feature_size = (2*numcontext+1)*numcep, batch_size multiplied by num_gpus) and the same dotted-name
network loader.  `network=networks.bilstm_ctc_net.BiLstmCTCNet` resolves to the HIP implementation in
neuralasr_amd.networks when the reference's TensorFlow package of that name is not importable."""
import importlib
from configparser import ConfigParser, ExtendedInterpolation

from .logger import get_logger
from .symbols import Symbols

log = get_logger()

_INT_KEYS = ('samplerate', 'numcep', 'batch_size', 'epochs', 'start_step', 'report_step', 'num_gpus', 'label_context')


class Config(object):
    def __init__(self, configfile, isTraining=False):
        self.isTraining = isTraining
        self.configfile = configfile
        log.info('Reading configuration from: ' + configfile)
        self.cfg = ConfigParser(interpolation=ExtendedInterpolation())
        self.cfg.read(configfile)
        par = self.cfg['Parameters']
        for key in _INT_KEYS:
            setattr(self, key, int(par[key]))
        self.numcontext = int(par['numcontext']) if 'numcontext' in par else 0
        self.rand_shift = int(par['rand_shift']) if 'rand_shift' in par else 0
        self.learningrate = float(par['learningrate'])
        self.model_dir = par['model_dir']
        self.punc_regex = par['punc_regex']
        self.network = par['network']
        self.sym_file = par['sym_file'] if 'sym_file' in par else None
        self.feature_size = (2 * self.numcontext + 1) * self.numcep
        # the configured batch is per GPU (reference: config.py:35-36)
        self.batch_size *= self.num_gpus if self.num_gpus > 0 else 1
        self.symbols = Symbols(self.label_context, self.sym_file) if isTraining else Symbols(self.label_context)

        train, test, feat = self.cfg['Train'], self.cfg['Test'], self.cfg['MFCC Featurizer']
        self.train_input = train['input'] if 'input' in train else None
        self.mfcc_input = feat['input'] if 'input' in feat else None
        self.mfcc_output = feat['output'] if 'output' in feat else None
        self.start_marker = feat['start_marker'] if 'start_marker' in feat else None
        self.end_marker = feat['end_marker'] if 'end_marker' in feat else None
        self.test_input = test['input'] if 'input' in test else None
        if self.test_input is None and not self.train_input:
            raise ValueError("Missing 'test_input' in configuration file: " + configfile)

    def load_network(self, fortraining=False):
        parts = self.network.split('.')
        modname, classname = '.'.join(parts[:-1]), parts[-1]
        module = None
        for candidate in (modname, 'neuralasr_amd.' + modname):
            try:
                module = importlib.import_module(candidate)
                getattr(module, classname)
                break
            except (ImportError, AttributeError):
                module = None
        if module is None:
            raise ImportError('cannot load network class ' + self.network)
        return getattr(module, classname)(self, fortraining=fortraining)

    def print_config(self):
        names = ['samplerate', 'numcep', 'numcontext', 'rand_shift', 'batch_size', 'epochs', 'learningrate',
                 'model_dir', 'start_step', 'report_step', 'num_gpus', 'label_context', 'punc_regex', 'network',
                 'sym_file', 'train_input', 'test_input', 'mfcc_input', 'mfcc_output', 'start_marker', 'end_marker']
        lines = ['']
        for n in names:
            v = getattr(self, n)
            lines.append(('%s=%f' % (n, v)) if n == 'learningrate' else '%s=%s' % (n, v))
        log.info('\n'.join(lines) + '\n')

    def write_symbols(self):
        self.symbols.write(self.sym_file)

    def write(self, filename):
        log.info('Writing configuration to: ' + filename)
        with open(filename, 'w') as fh:
            self.cfg.write(fh)
