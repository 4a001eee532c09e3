"""The per-model library exports exactly what include/egdst.h declares, and refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest

from egdst_amd import build, examples, runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    return build.build_model(examples.deaton1())   # hipcc cross-compiles gfx950 without a GPU


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'egdst.h')).read()
    declared = set(re.findall(r'^(?:const char \*|int |double )(egdst_[A-Za-z_]+)\(', hdr, flags=re.M))
    assert declared == set(runtime.ABI_SYMBOLS), declared ^ set(runtime.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib.lib, s)


def test_model_info_and_error_texts(lib):
    i = lib.info
    assert (i.nst, i.nd, i.nparam, i.neq, i.distrib) == (1, 1, 2, 1, 1)
    assert i.optim_MUnoD == 1 and i.optim_UnoD == 1 and abs(i.tolerance - 1e-10) < 1e-25
    assert b'Transition probabilities' in lib.lib.egdst_strerror(11)
    assert b'no CPU fallback' in lib.lib.egdst_strerror(3)


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    m = examples.deaton1()
    with pytest.raises(runtime.EgdstRuntimeError) as e:
        runtime.Solver(lib, m.descriptor(), ndraw=1)
    assert e.value.code in (2, 3)


def test_missing_library_is_an_error():
    with pytest.raises(runtime.EgdstRuntimeError):
        runtime.ModelLibrary('/nonexistent/libegdst.so')


def test_product_never_imports_the_oracle():
    """Nothing under egdst_amd/ may import, include, link or load anything from oracle/ or tests/."""
    pat_py = re.compile(r'^\s*(?:from|import)\s+(?:oracle|tests|oracle_harness|build_oracle|cpu_emu)|CDLL\([^)]*oracle|oracle/_build')
    pat_c = re.compile(r'#\s*include\s*"[^"]*(?:oracle|cpu_emu)')
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'egdst_amd')):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith('.py'):
                for line in open(path, errors='replace'):
                    assert not pat_py.search(line), (path, line)
            elif f.endswith(('.hip', '.h', '.inc')) and '_models' not in dirpath:
                for line in open(path, errors='replace'):
                    assert not pat_c.search(line), (path, line)
