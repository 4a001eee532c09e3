"""Build the device code of one model for the CPU sanitizer harness (ASan+UBSan, one-thread workgroups)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'egdst_amd', 'csrc')


def build(spec_dir, sanitize=True, env_bs=1, parallel_blocks=False, wave=1, fix_waves=2):
    """sanitize: True/'address' -> ASan+UBSan, 'thread' -> TSan (use env_bs>1), False -> none."""
    kind = 'address' if sanitize is True else sanitize
    out = os.path.join(spec_dir, 'libegdst_cpuemu_%s_%d_w%d_f%d%s.so' % (kind or 'plain', env_bs, wave, fix_waves, '_pb' if parallel_blocks else ''))
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(spec_dir, 'modelspec.h'), os.path.join(ROOT, 'include', 'egdst_math.h'),
                                                                  os.path.join(HERE, 'hip', 'hip_runtime.h')]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(s) for s in srcs):
        return out
    cmd = ['g++', '-x', 'c++', '-std=c++17', '-O1', '-g', '-mfma', '-ffp-contract=off', '-fPIC', '-shared',
           '-DEGDST_EMU', '-DENV_SEG_MINPTS=%d' % (4 * wave), '-DWAVE=%d' % wave, '-DGRID_BS=1', '-DENV_BS_EMU=%d' % env_bs, '-DFIX_BS=%d' % (1 if parallel_blocks else fix_waves * wave), '-pthread', '-I', HERE, '-I', spec_dir, '-I', CSRC, '-I', os.path.join(ROOT, 'include'),
           '-Wno-unused-function', os.path.join(CSRC, 'egdst_kernels.hip'), '-o', out]
    extra = os.environ.get('EMU_EXTRA_FLAGS', '').split()  # e.g. -DENV_SEGNF_SLICE=4: wide cursor slices with few pieces
    if extra:
        cmd[1:1] = extra
        out = out.replace('.so', '_' + ''.join(c if c.isalnum() else '_' for c in ''.join(extra)) + '.so')
        cmd[-1] = out
        if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(s) for s in srcs):
            return out
    if os.environ.get('EMU_SEQ_WALK'):
        cmd[1:1] = ['-DEGDST_SEQ_WALK']
        out = out.replace('.so', '_seqwalk.so')
        cmd[-1] = out
    if parallel_blocks:
        cmd[1:1] = ['-DEMU_PARALLEL_BLOCKS']
    if kind == 'address':
        cmd[1:1] = ['-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-fno-sanitize-recover=undefined']
    elif kind == 'thread':
        cmd[1:1] = ['-fsanitize=thread', '-fno-omit-frame-pointer']
    subprocess.run(cmd, check=True)
    return out


if __name__ == '__main__':
    print(build(sys.argv[1]))
