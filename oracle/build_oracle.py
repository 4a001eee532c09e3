"""Build the CPU oracle for one generated model plugin (TEST INFRASTRUCTURE).

    python oracle/build_oracle.py <dir containing modelspec.h> [out.so]

gcc -O2 -ffp-contract=off (no FMA contraction, no -ffast-math): SURVEY.md §8(d) "CPU baseline beside it".
-mfma only turns the explicit fma() calls of include/egdst_math.h into one instruction (results are the same without).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(spec_dir, out=None, opt='-O2', native_math=True):
    """native_math=True: the platform libm (glibc exp/log/pow, what the reference MEX runs on); False: include/egdst_math.h
    (the same algorithm restated, which the GPU runs too; equal to glibc bit for bit)."""
    out = out or os.path.join(spec_dir, 'liboracle_%s.so' % ('native' if native_math else 'portable'))
    src = os.path.join(HERE, 'egdst_oracle.c')
    spec = os.path.join(spec_dir, 'modelspec.h')
    inc = os.path.join(os.path.dirname(HERE), 'include')
    if (os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(spec),
                                                                os.path.getmtime(os.path.join(inc, 'egdst_math.h')),
                                                                os.path.getmtime(os.path.join(inc, 'egdst_math_tables.h')))):
        return out
    cmd = ['gcc', opt, '-mfma', '-ffp-contract=off', '-std=gnu99', '-fPIC', '-shared', '-Wall', '-Wno-unused-function',
           '-Wno-unused-variable', '-Wno-unused-but-set-variable', '-I', spec_dir, '-I', inc] + (['-DEGDST_NATIVE_MATH'] if native_math else []) + [src, '-o', out, '-lm']
    subprocess.run(cmd, check=True)
    return out


if __name__ == '__main__':
    print(build(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None))
