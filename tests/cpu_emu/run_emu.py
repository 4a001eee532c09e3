"""Run one example through the sanitizer build and compare with the oracle (CPU only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'cpu_emu'))
import build_emu  # noqa: E402
from egdst_amd import build, codegen, runtime  # noqa: E402
from oracle_harness import Oracle  # noqa: E402
from parity import compare  # noqa: E402


def run(model, sanitize=True, env_bs=1, parallel_blocks=False, wave=1):
    text = codegen.generate_modelspec(model)
    d = os.path.join(build.MODELS_DIR, build.model_tag(model, text))
    os.makedirs(d, exist_ok=True)
    spec = os.path.join(d, 'modelspec.h')
    if not os.path.exists(spec) or open(spec).read() != text:
        open(spec, 'w').write(text)
    lib = runtime.ModelLibrary(build_emu.build(d, sanitize, env_bs, parallel_blocks, wave))
    s = runtime.Solver(lib, model.descriptor(), ndraw=1, keep_history=True, rows_cap=int(os.environ.get('EMU_ROWS_CAP', '0')))
    s.set_params(model.param_vector())
    rc = s.solve(raise_on_error=False)
    sol = s.solution(0)
    ref = Oracle(model).solve()
    ok, rep = compare(sol, ref)
    return ok, rep, sol, ref, s


if __name__ == '__main__':
    from egdst_amd import examples
    name = sys.argv[1]
    kw = eval('dict(%s)' % (sys.argv[2] if len(sys.argv) > 2 else ''))
    san = os.environ.get('EMU_SANITIZE', 'address')
    ok, rep, sol, ref, s = run(examples.REGISTRY[name](**kw), {'0': False}.get(san, san), int(os.environ.get('EMU_ENV_BS', '1')),
                                bool(int(os.environ.get('EMU_PAR_BLOCKS', '0'))), int(os.environ.get('EMU_WAVE', '1')))
    print(name, 'ok=%s status=%d where=%s rows=%d/%d evals=%d/%d max_rel=%.2e max_dth=%.2e %s' % (
        ok, sol.status, sol.where, sol.total_rows(), ref.total_rows(), sol.nevals, ref.nevals, rep['max_rel'],
        rep['max_dth'], rep['problems'][:3]), 'geometry', s.geometry(), 'capacity_retries', s.capacity_retries, 'deferrals', int(s.debug(0)[15]), 'walks segmented/fallback', s.walk_stats()[0].tolist(), 'tp done/left', s.tp_stats()[0].tolist())
