"""Builds neuralasr_amd/libnasr.so (the HIP kernels + C ABI of include/nasr.h) for gfx950 with hipcc.
In-tree, explicit `hipcc -shared -fPIC`; cross-compiles without a GPU.  `python -m neuralasr_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libnasr.so')
SOURCES = ['gemm.hip', 'gemm_tph.hip', 'lstm.hip', 'lstm_persist.hip', 'lstm_wide.hip', 'ctc.hip', 'dense.hip', 'optim.hip', 'nasr_layout.hip',
           'nasr_batch.hip', 'nasr_pass.hip', 'nasr_api.hip', 'nasr_comm.hip', 'beam.cpp']
HEADERS = [os.path.join(CSRC, 'kernels.h'), os.path.join(CSRC, 'nasr_ctx.h'), os.path.join(HERE, '..', 'include', 'nasr.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    hipcc = _hipcc()
    bdir = os.path.join(HERE, 'build')
    os.makedirs(bdir, exist_ok=True)
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(bdir, os.path.splitext(s)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-lpthread'])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
