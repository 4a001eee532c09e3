"""Per-queue view of the last solve in a rocprofv3 kernel trace csv: for every HIP stream (queue) of a batched solve the busy time
(sum of kernel durations), the span, and the busy time by kernel (diagnostics: python tools/trace_queues.py DIR [NSOLVES])."""
import csv, collections, glob, sys
d = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) - len(rows) // N:]
t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
print('last solve: %d kernels, span %.1f ms; columns: %s' % (len(rows), (t1 - t0) / 1e6, [k for k in rows[0].keys() if 'ueue' in k or 'tream' in k]))
qk = 'Queue_Id' if 'Queue_Id' in rows[0] else [k for k in rows[0].keys() if 'ueue' in k][0]
by = collections.defaultdict(list)
for r in rows:
    by[r[qk]].append(r)
names = sorted({r['Kernel_Name'].split('(')[0] for r in rows})
print('%-8s %6s %8s %8s  ' % ('queue', 'n', 'busy ms', 'span ms') + ' '.join('%11s' % n[:11] for n in names))
tot = collections.defaultdict(float)
for q, rs in sorted(by.items(), key=lambda kv: -sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in kv[1])):
    busy = collections.defaultdict(float)
    for r in rs:
        busy[r['Kernel_Name'].split('(')[0]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    span = (max(int(r['End_Timestamp']) for r in rs) - min(int(r['Start_Timestamp']) for r in rs)) / 1e6
    print('%-8s %6d %8.1f %8.1f  ' % (q, len(rs), sum(busy.values()), span) + ' '.join('%11.1f' % busy[n] for n in names))
    for n in names:
        tot[n] += busy[n]
print('%-8s %6s %8.1f %8s  ' % ('sum', '', sum(tot.values()), '') + ' '.join('%11.1f' % tot[n] for n in names))
