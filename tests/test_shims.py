"""The MEX shim sources (shims/*.c: the reference-side bindings of the three gateways, egdst_solver.c:143, egdst_simulator.c:47,
egdst_call.c:17) are held to the MEX C API's types and to include/egdst.h by the compiler: gcc -fsyntax-only -Wall -Wextra
-Werror against a DECLARATIONS-ONLY mex.h (tests/mex_decls).  An undefined identifier, a wrong argument count or a pointer
mismatch in a shim fails here.  No MATLAB exists in the image, so nothing is linked; the sequence of library calls the shims
make (create, set_params, set_cell_M/D, simulate / call, destroy) runs on the GPU in tests/test_gpu_parity.py::test_import_*."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = sorted(glob.glob(os.path.join(ROOT, 'shims', '*.c')))


def test_there_is_one_shim_per_gateway():
    assert [os.path.basename(p) for p in SHIMS] == ['egdst_call_hip.c', 'egdst_simulator_hip.c', 'egdst_solver_hip.c']


@pytest.mark.parametrize('src', SHIMS, ids=[os.path.basename(p) for p in SHIMS])
@pytest.mark.parametrize('cc', ['gcc -std=c99', 'g++ -x c++ -std=c++17'])
def test_shim_compiles_against_the_mex_api_and_the_c_abi(src, cc):
    cmd = cc.split() + ['-fsyntax-only', '-Wall', '-Wextra', '-Werror', '-I', os.path.join(ROOT, 'include'),
                        '-I', os.path.join(ROOT, 'shims'), '-I', os.path.join(ROOT, 'tests', 'mex_decls'), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_mex_decls_header_has_no_function_bodies():
    """declarations only: the header cannot make anything link or run"""
    text = open(os.path.join(ROOT, 'tests', 'mex_decls', 'mex.h')).read()
    code = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    assert '{' not in code.replace('extern "C" {', '').replace('typedef enum {', '')


def test_shims_keep_no_state_between_calls():
    """every gateway call creates and destroys its handle: no statics, no persistent handles, no mexLock"""
    for p in SHIMS + [os.path.join(ROOT, 'shims', 'egdst_shim_common.h')]:
        code = re.sub(r'/\*.*?\*/', '', open(p).read(), flags=re.S)
        assert 'mexLock' not in code and 'mexAtExit' not in code, p
        for line in code.splitlines():
            if re.match(r'\s*static\s', line):
                assert '(' in line, (p, line)   # static functions only, no static variables
        if p.endswith('.c'):
            assert code.count('egdst_destroy(h)') >= 1 and 'shim_handle(' in code, p


def test_every_library_call_of_the_shims_is_declared_in_the_header():
    hdr = open(os.path.join(ROOT, 'include', 'egdst.h')).read()
    declared = set(re.findall(r'^(?:const char \*|int |double )(egdst_[A-Za-z_]+)\(', hdr, flags=re.M))
    used = set()
    for p in SHIMS + [os.path.join(ROOT, 'shims', 'egdst_shim_common.h')]:
        code = re.sub(r'/\*.*?\*/', '', open(p).read(), flags=re.S)
        used |= set(re.findall(r'\b(egdst_[a-z_A-Z]+)\(', code))
    assert used <= declared, used - declared
    assert {'egdst_set_cell_M', 'egdst_set_cell_D', 'egdst_simulate', 'egdst_call', 'egdst_get_dbgout'} <= used
