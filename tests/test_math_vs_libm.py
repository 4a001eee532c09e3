"""include/egdst_math.h restates glibc's exp/log/pow (x86-64 FMA build) operation by operation; this test holds it to
that: 0 ulp against the platform libm over millions of arguments per function, with and without hardware FMA in the
test binary (the explicit fma() calls are correctly rounded either way).  The GPU runs the same header
(tests/test_gpu_parity.py::test_device_math_equals_host_libm)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _glibc_fma_variant():
    """The claim is about glibc >= 2.28 on an x86-64 host whose libm dispatches to the FMA variants."""
    try:
        flags = open('/proc/cpuinfo').read()
    except OSError:
        return False
    return ' fma ' in flags and ' avx2 ' in flags and sys.platform.startswith('linux')


@pytest.mark.skipif(not _glibc_fma_variant(), reason='host libm is not the glibc FMA build')
@pytest.mark.parametrize('mfma', [True, False])
def test_exp_log_pow_equal_glibc_bit_for_bit(tmp_path, mfma):
    exe = str(tmp_path / 'math_check')
    cmd = ['gcc', '-O2', '-ffp-contract=off', '-I', os.path.join(ROOT, 'include'), os.path.join(HERE, 'math_check.c'),
           '-o', exe, '-lm'] + (['-mfma'] if mfma else [])
    subprocess.run(cmd, check=True)
    n = 4000000 if mfma else 1000000
    r = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert 'mismatches exp=0 log=0 pow=0' in r.stdout, r.stdout[-2000:]
