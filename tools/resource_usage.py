#!/usr/bin/env python
"""Register / scratch / LDS usage of every kernel of a model library, as hipcc reports it (-Rpass-analysis=kernel-resource-usage).

    python tools/resource_usage.py C2 [-DENV_MINW=3 ...]  > profiles/r03_resource_usage_C2.txt

Cross-compiles for gfx950 without a GPU; nothing is run."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from egdst_amd import build, codegen, workloads  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'C2'
    extra = sys.argv[2:]
    model = workloads.WORKLOADS[wl]()[0]
    text = codegen.generate_modelspec(model)
    d = os.path.join(build.MODELS_DIR, build.model_tag(model, text))
    os.makedirs(d, exist_ok=True)
    spec = os.path.join(d, 'modelspec.h')
    if not os.path.exists(spec) or open(spec).read() != text:
        open(spec, 'w').write(text)
    cmd = [build._hipcc()] + [f for f in build.HIPCC_FLAGS if f not in ('-shared',)] + extra + [
        '-Rpass-analysis=kernel-resource-usage', '-c', '-I', d, '-I', build.CSRC, '-I', os.path.join(ROOT, 'include'),
        os.path.join(build.CSRC, 'egdst_kernels.hip'), '-o', '/dev/null']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-4000:])
    rows, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', ln)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r'remark:\s+([A-Za-z0-9 \[\]/]+): (\S+)', ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    print('# %s %s: hipcc %s' % (wl, ' '.join(extra), ' '.join(build.HIPCC_FLAGS)))
    print('%-36s %6s %6s %6s %10s %10s %10s %8s %6s' % ('kernel', 'VGPRs', 'AGPRs', 'SGPRs', 'VGPRspill', 'SGPRspill', 'scratch[B]', 'LDS[B]', 'occ'))
    for c in rows:
        nm = subprocess.run(['c++filt', c['name']], capture_output=True, text=True).stdout.strip().split('(')[0]
        print('%-36s %6s %6s %6s %10s %10s %10s %8s %6s' % (nm[:36], c.get('VGPRs', '?'), c.get('AGPRs', '?'), c.get('TotalSGPRs', '?'),
              c.get('VGPRs Spill', '?'), c.get('SGPRs Spill', '?'), c.get('ScratchSize [bytes/lane]', '?'),
              c.get('LDS Size [bytes/block]', '?'), c.get('Occupancy [waves/SIMD]', '?')))


if __name__ == '__main__':
    main()
