"""profiles/pmc_metrics.json from the per-kernel counter summaries of tools/pmc_run.sh (profiles/rNN_*_pmc_<WL>_ndraw<N>.csv).

    python tools/pmc_to_json.py profiles/r02_a_pmc_C2_ndraw4096.csv [more.csv ...]

Per kernel (all figures are of the PMC passes themselves, in which rocprofv3 serialises the dispatches):
  hbm_bytes_per_launch      (FETCH_SIZE + WRITE_SIZE) KB * 1024 / dispatches, as reported (MI355X_MICROARCH.md: FETCH_SIZE
                            halves 16-B-per-lane streaming reads on gfx950; this path reads 8 B per lane, uncalibrated: not doubled)
  valu_util                 SQ_ACTIVE_INST_VALU * 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
  valu_fp64_util            (ADD+MUL+FMA+TRANS)_F64 wave instructions * 4 cycles (16 fp64 lanes per SIMD and cycle) / the same
  lds_GBps                  SQ_INSTS_LDS * 64 lanes * 8 B / kernel busy time (an upper estimate: 8 B per lane per instruction)
  wave_cycles_waiting_frac  SQ_WAIT_ANY / SQ_WAVE_CYCLES
  icache_miss_rate          SQC_ICACHE_MISSES / SQC_ICACHE_REQ
  l2_hit_rate               TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_HZ = 2.4e9  # peak engine clock; GRBM_GUI_ACTIVE counts cycles, busy time = cycles / 8 XCDs / clock (lower bound of time)


def main():
    out = {}
    for f in sys.argv[1:]:
        m = re.search(r'pmc_(C\d)_ndraw(\d+)', f)
        key = '%s_ndraw%s' % (m.group(1), m.group(2))
        c = defaultdict(dict)
        n = {}
        for row in csv.DictReader(open(f)):
            c[row['kernel']][row['counter']] = float(row['sum'])
            n[row['kernel']] = int(row['dispatches'])
        cfg = {}
        for k, v in c.items():
            if 'GRBM_GUI_ACTIVE' not in v:
                continue
            cyc = v['GRBM_GUI_ACTIVE'] / 8.0
            simd = cyc * 1024.0
            f64 = sum(v.get('SQ_INSTS_VALU_%s_F64' % t, 0.0) for t in ('ADD', 'MUL', 'FMA', 'TRANS'))
            cfg[k] = {
                'dispatches': n[k],
                'hbm_bytes_per_launch': (v.get('FETCH_SIZE', 0.0) + v.get('WRITE_SIZE', 0.0)) * 1024.0 / n[k],
                'valu_util': v.get('SQ_ACTIVE_INST_VALU', 0.0) * 4.0 / simd,
                'valu_fp64_util': f64 * 4.0 / simd,
                'lds_GBps': v.get('SQ_INSTS_LDS', 0.0) * 64 * 8 / (cyc / CLOCK_HZ) / 1e9,
                'wave_cycles_waiting_frac': v.get('SQ_WAIT_ANY', 0.0) / max(v.get('SQ_WAVE_CYCLES', 1.0), 1.0),
                'icache_miss_rate': v.get('SQC_ICACHE_MISSES', 0.0) / max(v.get('SQC_ICACHE_REQ', 1.0), 1.0),
                'l2_hit_rate': v.get('TCC_HIT_sum', 0.0) / max(v.get('TCC_HIT_sum', 0.0) + v.get('TCC_MISS_sum', 0.0), 1.0),
                'source': os.path.relpath(os.path.abspath(f), ROOT),
            }
        out[key] = cfg
    path = os.path.join(ROOT, 'profiles', 'pmc_metrics.json')
    json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
    print('wrote', path)
    for key, cfg in out.items():
        for k, v in sorted(cfg.items()):
            print(key, k, {a: (round(b, 5) if isinstance(b, float) else b) for a, b in v.items() if a != 'source'})


if __name__ == '__main__':
    main()
