"""Output-symbol table (reference: symbols.py:8-68): sym <-> id, `<padding>` inserted first (id 0),
`<blank>` last (= num_classes-1, TF's CTC blank); file format "sym id" per line, sorted by sym."""
import os

from .logger import get_logger

log = get_logger()


class Symbols(object):
    blank = '<blank>'
    padding = '<padding>'

    def __init__(self, label_context, filename=None):
        self.label_context = label_context
        self.filename = filename
        self.sym_to_id = {}
        self.id_2_sym = {}
        self.counter = 0
        if filename and os.path.exists(filename):
            log.info('Reading output symbols from: ' + filename)
            top = 0
            with open(filename, 'r') as fh:
                for line in fh:
                    parts = line.strip().split(' ')
                    sym, idx = parts[0], int(parts[1])
                    self.sym_to_id[sym] = idx
                    top = max(top, idx)
            self.id_2_sym = {v: k for k, v in self.sym_to_id.items()}
            self.counter = top + 1

    # -- insertion
    def insert_sym(self, sym):
        idx = self.sym_to_id.get(sym)
        if idx is None:
            idx = self.counter
            self.sym_to_id[sym] = idx
            self.id_2_sym[idx] = sym
            self.counter += 1
        return idx

    def insert_blank(self):
        return self.insert_sym(self.blank)

    def insert_padding(self):
        return self.insert_sym(self.padding)

    # -- lookup
    def get_padding_id(self):
        return self.sym_to_id[self.padding]

    def get_id(self, sym):
        return self.sym_to_id[sym]

    def get_sym(self, id):
        return self.id_2_sym[id]

    def get_all_ids(self, id=None):
        return list(self.sym_to_id.values())

    def convert_to_str(self, ids):
        ctx = self.label_context
        if ctx > 0:
            pieces = [self.get_sym(i)[ctx:-ctx] for i in ids]
        else:
            pieces = [self.get_sym(i) for i in ids]
        return ''.join(pieces).replace(self.blank, '').replace('_', ' ')

    def write(self, filename=None):
        filename = filename or self.filename
        log.info('Writing output symbols to: ' + filename)
        with open(filename, 'w') as fh:
            for sym in sorted(self.sym_to_id):
                fh.write('%s %d\n' % (sym, self.sym_to_id[sym]))
