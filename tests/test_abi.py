"""The C-ABI library loads and exports every symbol include/nasr.h declares (no compute calls: the dev
container has no GPU), and the product path fails loudly — no CPU fallback — when no GPU is usable."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'nasr.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nasr_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from neuralasr_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/nasr.h but not exported by libnasr.so'
    assert sorted(_lib.SYMBOLS) == names, 'ctypes binding and header disagree'


def test_structs_match_header_layout():
    from neuralasr_amd import _lib
    assert ctypes.sizeof(_lib.ModelCfg) == 6 * 4 + 5 * 4 + (1 + 3 + 1) * 4 + (1 + 4) * 4   # + the dense-stage fields
    assert ctypes.sizeof(_lib.PhaseTimes) == 9 * 4 + 2 * 4


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from neuralasr_amd import _lib
    from neuralasr_amd.engine import Engine
    with pytest.raises(_lib.NasrError, match='no HIP device|no CPU fallback'):
        Engine(8, 16, 1, True, 'stack_reshape', 5)


def test_product_package_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/ (comments may cite it)."""
    pkg = os.path.join(ROOT, 'neuralasr_amd')
    py_use = re.compile(r'^\s*(from\s+oracle|import\s+oracle|from\s+\.+oracle)|oracle\.nasr_oracle|nasr_oracle\s*\(', re.M)
    c_use = re.compile(r'#\s*include\s*[<"][^>"]*oracle', re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            src_path = os.path.join(dirpath, f)
            if f.endswith('.py'):
                assert not py_use.search(open(src_path).read()), f'{f} imports the oracle'
            elif f.endswith(('.hip', '.h', '.cpp')):
                assert not c_use.search(open(src_path).read()), f'{f} includes oracle code'
