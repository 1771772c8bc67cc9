#!/usr/bin/env python3
"""Folds the in-kernel phase stamps tools/persistbench prints (a -DNASR_PSTAMP=1 build; wave 0 of every workgroup, cycles per
timestep, median over the 256 workgroups of the last repetition) into profiles/persist_stamps.json, which bench.py quotes
as roofline.latency.phase_cycles.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNASR_PSTAMP=1 tools/persistbench.hip neuralasr_amd/csrc/lstm.hip \
          neuralasr_amd/csrc/lstm_persist.hip -o tools/sb_st && tools/sb_st 500 16 500 2 0 > log
    python tools/stamps_to_json.py <workload key> log [profiles/persist_stamps.json]"""
import json
import os
import re
import sys


def main():
    key, log = sys.argv[1:3]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), '..', 'profiles', 'persist_stamps.json')
    fwd, bwd, step = {}, {}, {}
    for line in open(log):
        m = re.match(r'\s+(fwd|bwd) (\S+)\s+over 256 CUs: min (\d+)\s+p10 (\d+)\s+median (\d+)\s+p90 (\d+)\s+max (\d+)', line)
        if m:
            (fwd if m.group(1) == 'fwd' else bwd)[m.group(2)] = {'min': int(m.group(3)), 'median': int(m.group(5)), 'max': int(m.group(7))}
        m = re.match(r'(forward |backward): per-step [\d.]+ us/step\s+persistent ([\d.]+) us/step', line)
        if m:
            step['fwd_us_stamped' if m.group(1).startswith('forward') else 'bwd_us_stamped'] = float(m.group(2))
    doc = json.load(open(out)) if os.path.exists(out) else {}
    doc['_note'] = ('cycles per timestep of wave 0, per phase, median / min / max over the 256 workgroups (s_memtime stamps, '
                    '-DNASR_PSTAMP=1 build of tools/persistbench at 2x500, B 16, T 500; the stamps themselves add ~10 % to the '
                    'step).  tools/stamps_to_json.py')
    doc[key] = {'forward': fwd, 'bptt': bwd, **step}
    json.dump(doc, open(out, 'w'), indent=1)
    print(json.dumps(doc[key]))


if __name__ == '__main__':
    main()
