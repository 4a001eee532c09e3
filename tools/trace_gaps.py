"""Gaps and concurrency in a rocprofv3 kernel trace csv of batched solves (diagnostics: python tools/trace_gaps.py DIR [NSOLVES]).

For the last solve in the trace: per HIP stream (queue) the gaps between consecutive kernels -- the next kernel of a stream depends on
the one before it, so a gap is time the stream's chain spends neither running nor waiting for CUs inside a kernel -- by the pair
(kernel before, kernel after); and over the whole solve the number of kernels in flight (time-weighted histogram)."""
import csv, collections, glob, sys
d = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) - len(rows) // N:]
qk = 'Queue_Id' if 'Queue_Id' in rows[0] else [k for k in rows[0].keys() if 'ueue' in k][0]
t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
span = (t1 - t0) / 1e6
by = collections.defaultdict(list)
for r in rows:
    by[r[qk]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
gap = collections.defaultdict(list)
busy_q, gap_q = [], []
for q, rs in by.items():
    rs.sort()
    busy_q.append(sum(e - s for s, e, _ in rs) / 1e6)
    g = 0.0
    for (s0, e0, n0), (s1, e1, n1) in zip(rs, rs[1:]):
        gap[(n0, n1)].append((s1 - e0) / 1e3)
        g += max(0, s1 - e0) / 1e6
    gap_q.append(g)
print('last solve: %d kernels on %d queues, span %.1f ms' % (len(rows), len(by), span))
print('per queue: kernels busy %.1f ms (min %.1f max %.1f), gaps between its kernels %.1f ms (min %.1f max %.1f)' % (
    sum(busy_q) / len(busy_q), min(busy_q), max(busy_q), sum(gap_q) / len(gap_q), min(gap_q), max(gap_q)))
print('%-30s %7s %9s %9s %9s %10s' % ('gap after -> before', 'n', 'median us', 'mean us', 'p90 us', 'sum ms/q'))
for (a, b), v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('%-30s %7d %9.1f %9.1f %9.1f %10.2f' % (a[:14] + ' -> ' + b[:12], len(v), v[len(v) // 2], sum(v) / len(v), v[int(len(v) * 0.9)], sum(v) / 1e3 / len(by)))
# kernels in flight
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
hist = collections.defaultdict(float)
cur, last = 0, ev[0][0]
for t, dlt in ev:
    hist[cur] += t - last
    cur += dlt; last = t
tot = sum(hist.values())
print('kernels in flight (share of the span): ' + '  '.join('%d: %.1f%%' % (k, 100 * v / tot) for k, v in sorted(hist.items())))
print('mean kernels in flight %.2f' % (sum(k * v for k, v in hist.items()) / tot))
