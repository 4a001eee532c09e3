"""Per-model build of the HIP library: the `mex` step of compile.m:754-819 for gfx950.

The generated plugin (modelspec.h) is compiled together with csrc/egdst_kernels.hip into
``egdst_amd/_models/<label>_<hash>/libegdst.so`` with hipcc for --offload-arch=gfx950.  The
library is kept IN-TREE (git-ignored) so that it travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

from . import codegen

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
MODELS_DIR = os.path.join(HERE, '_models')
SOURCES = ['egdst_kernels.hip', 'egdst_device.h', 'egdst_envelope.h', 'egdst_host.inc']
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-std=c++17', '-fPIC', '-shared',
               '-Wno-unused-value', '-Wno-unused-parameter']


class BuildError(RuntimeError):
    pass


def model_tag(model, text=None):
    text = text if text is not None else codegen.generate_modelspec(model)
    label = ''.join(ch for ch in model.label if ch.isalnum())[:16] or 'model'
    return '%s_%s' % (label, codegen.spec_hash(text))


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise BuildError('hipcc not found: the egdst hot path has no CPU fallback and cannot be built')


def build_model(model, build_dir=None, force=False, extra_flags=()):
    """Generate modelspec.h and build libegdst.so for `model`; returns runtime.ModelLibrary."""
    from . import runtime
    text = codegen.generate_modelspec(model)
    tag = model_tag(model, text)
    # build variants (extra -D flags from the caller -- workloads.BATCH_BUILD_FLAGS -- or, for experiments, from the
    # environment) live in directories of their own
    extra_flags = list(extra_flags) + os.environ.get('EGDST_HIPCC_EXTRA', '').split()
    if extra_flags and not build_dir:
        tag += '_' + ''.join(c if c.isalnum() else '_' for c in ''.join(extra_flags))
    d = build_dir or os.path.join(MODELS_DIR, tag)
    os.makedirs(d, exist_ok=True)
    spec = os.path.join(d, 'modelspec.h')
    if not os.path.exists(spec) or open(spec).read() != text:
        with open(spec, 'w') as f:
            f.write(text)
    lib = os.path.join(d, 'libegdst.so')
    newest = max(os.path.getmtime(os.path.join(CSRC, s)) for s in SOURCES)
    newest = max(newest, os.path.getmtime(spec), os.path.getmtime(os.path.join(HERE, '..', 'include', 'egdst.h')),
                 os.path.getmtime(os.path.join(HERE, '..', 'include', 'egdst_math.h')),
                 os.path.getmtime(os.path.join(HERE, '..', 'include', 'egdst_math_tables.h')))
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        tmp = '%s.tmp.%d' % (lib, os.getpid())  # (several ranks may find the library stale at once: build aside, rename)
        cmd = [_hipcc()] + HIPCC_FLAGS + list(extra_flags) + ['-I', d, '-I', CSRC, '-I', os.path.join(HERE, '..', 'include'),
                                                              os.path.join(CSRC, 'egdst_kernels.hip'), '-o', tmp]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise BuildError('hipcc failed for model %r:\n%s\n%s' % (model.label, ' '.join(cmd), r.stderr[-6000:]))
        os.replace(tmp, lib)
    model.make_simlabels()
    model.dir = d
    return runtime.ModelLibrary(lib, tag)
