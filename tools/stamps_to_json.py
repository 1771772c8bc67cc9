#!/usr/bin/env python3
"""Folds the in-kernel phase stamps tools/persistbench prints (a -DNASR_PSTAMP=1 build; wave 0 of every workgroup, cycles per
timestep, median over the 256 workgroups of the last repetition) into profiles/persist_stamps.json, which bench.py quotes
as roofline.latency.phase_cycles.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNASR_PSTAMP=1 tools/persistbench.hip neuralasr_amd/csrc/lstm.hip \
          neuralasr_amd/csrc/lstm_persist.hip -o tools/sb_st && tools/sb_st 500 16 500 2 0 > log
    python tools/stamps_to_json.py <workload key> log [profiles/persist_stamps.json]
    python tools/stamps_to_json.py --wide deepspeech widebench.log [profiles/persist_stamps.json]"""
import json
import os
import re
import sys


def wide(key, log, out):
    """tools/widebench (-DNASR_WSTAMP=1, WIDE_STAMPS=1): per phase 16 columns = the 8 waves of workgroups (0,0) and (7,31);
    waves 0 and 15 hold those workgroups' own column group (no exchange) and are left out of the statistics."""
    sec, doc_k, step = None, {'forward': {}, 'bptt': {}}, {}
    for line in open(log):
        if line.startswith('cycles per step'):
            sec = 'forward'
        elif line.startswith('BPTT cycles per step'):
            sec = 'bptt'
        m = re.match(r'\s+(\S+)\s+((?:[\d.]+\s*){16})$', line)
        if m and sec:
            v = sorted(float(x) for x in m.group(2).split()[1:15])
            doc_k[sec][m.group(1)] = {'min': round(v[0]), 'median': round(v[len(v) // 2]), 'max': round(v[-1])}
        m = re.match(r'(BPTT )?T \d+ B \d+: per-step kernels [\d.]+ ms \(([\d.]+) us/step\), wide persistent [\d.]+ ms \(([\d.]+) us per', line)
        if m:
            step['bwd_us_stamped' if m.group(1) else 'fwd_us_stamped'] = float(m.group(3))
    doc = json.load(open(out)) if os.path.exists(out) else {}
    doc[key] = {'_note': 'wide persistent kernels (lstm_wide.hip), cycles per DIRECTION-step of 14 waves of two workgroups '
                         '(tools/widebench, -DNASR_WSTAMP=1, Hp 2048, B 32, T 500)', **doc_k, **step}
    json.dump(doc, open(out, 'w'), indent=1)
    print(json.dumps(doc[key]))


def main():
    if sys.argv[1] == '--wide':
        out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(__file__), '..', 'profiles', 'persist_stamps.json')
        return wide(sys.argv[2], sys.argv[3], out)
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
