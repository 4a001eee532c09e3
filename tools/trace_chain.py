"""Where the wall time of a dependent kernel chain goes: busy time per kernel and the gaps between consecutive kernels, over the
last 1/N of a rocprofv3 kernel trace csv (diagnostics: python tools/trace_chain.py DIR [N])."""
import csv, collections, glob, sys
d = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) - len(rows) // N:]
busy = collections.defaultdict(float); cnt = collections.Counter(); gap_after = collections.defaultdict(float)
t_end = None; gaps = 0.0; last = None
for r in rows:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    k = r['Kernel_Name'].split('(')[0]
    busy[k] += (b - a) / 1e3; cnt[k] += 1
    if t_end is not None and a > t_end:
        gaps += (a - t_end) / 1e3; gap_after[last] += (a - t_end) / 1e3
    t_end = max(t_end or 0, b); last = k
wall = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
print('%d kernels, wall %.0f us, busy %.0f us, gaps %.0f us' % (len(rows), wall, sum(busy.values()), gaps))
for k in sorted(busy, key=lambda k: -busy[k]):
    print('  %-24s n=%5d  busy %8.0f us (%.1f each)   gap after it %8.0f us (%.1f each)' % (k, cnt[k], busy[k], busy[k] / cnt[k], gap_after[k], gap_after[k] / cnt[k]))
