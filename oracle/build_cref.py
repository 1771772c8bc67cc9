"""Builds oracle/build/libnasr_cref.so from oracle/cref/nasr_cref.c with gcc (x86-64-v3: AVX2+FMA, no -march=native
because the .so built in the dev container also runs on the GPU box's host CPU).  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'cref', 'nasr_cref.c')
OUT = os.path.join(HERE, 'build', 'libnasr_cref.so')


def build(force=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ['gcc', '-O3', '-mavx2', '-mfma', '-fopenmp', '-shared', '-fPIC', '-std=c11', '-o', OUT, SRC, '-lm']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('gcc failed:\n' + ' '.join(cmd) + '\n' + r.stderr)
    return OUT


if __name__ == '__main__':
    print(build(force=True))
