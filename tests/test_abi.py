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


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    """bench.py builds its workload, synthetic batch and weights itself; `oracle` appears only inside cpu_baseline()
    and the Workload.oracle_spec() helper that leg calls."""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, 'bench.py')).read())
    allowed = {'cpu_baseline', 'oracle_spec'}

    def visit(node, inside):
        for child in ast.iter_child_nodes(node):
            name = child.name if isinstance(child, (ast.FunctionDef, ast.ClassDef)) else None
            now = inside or (name in allowed)
            if isinstance(child, (ast.Import, ast.ImportFrom)):
                mods = [a.name for a in child.names] if isinstance(child, ast.Import) else [child.module or '']
                if any(m.split('.')[0] == 'oracle' for m in mods):
                    assert inside, f'bench.py imports the oracle outside its cpu_baseline leg (line {child.lineno})'
            visit(child, now)
    visit(tree, False)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/nasr.h compiles as C99, the struct layouts a C caller sees are the ones the ctypes binding assumes, and a C
    program linked against libnasr.so gets an error code (not a crash) from nasr_create when no GPU is usable."""
    import shutil
    import subprocess
    from neuralasr_amd import _lib
    gcc = shutil.which('gcc')
    if not gcc:
        pytest.skip('no gcc')
    src = tmp_path / 'abi.c'
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "nasr.h"
int main(void) {
  nasr_model_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.feature_size = 8; cfg.hidden = 16; cfg.num_layers = 1; cfg.bidirectional = 1; cfg.merge = NASR_MERGE_STACK_RESHAPE;
  cfg.num_classes = 5; cfg.forget_bias = 1.0f; cfg.learning_rate = 1e-3f; cfg.beta1 = 0.9f; cfg.beta2 = 0.999f; cfg.epsilon = 1e-8f;
  nasr_handle h = 0;
  int rc = nasr_create(&cfg, 0, 0, &h);
  printf("%zu %zu %d %d\n", sizeof(nasr_model_cfg), sizeof(nasr_phase_times), rc, h != 0);
  if (h) nasr_destroy(h);
  return 0;
}
''')
    exe = tmp_path / 'abi'
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [gcc, '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe),
           '-L', libdir, '-l:libnasr.so', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cfg_size, pt_size, rc, has_handle = (int(x) for x in out.stdout.split())
    assert cfg_size == ctypes.sizeof(_lib.ModelCfg) and pt_size == ctypes.sizeof(_lib.PhaseTimes)
    import torch
    if not torch.cuda.is_available():
        assert rc != 0 and not has_handle          # NASR_ERR_HIP: no device, no CPU fallback
    else:
        assert rc == 0 and has_handle
