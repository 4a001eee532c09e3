"""Median / mean duration per (kernel, grid size) of the second half of a rocprofv3 kernel trace csv (diagnostics)."""
import csv, collections, glob, sys
import numpy as np
for d in sys.argv[1:]:
    f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_')]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    half = rows[len(rows) // 2:]
    by = collections.defaultdict(list)
    for r in half:
        by[r['Kernel_Name'].split('(')[0] + ':' + r['Grid_Size_X']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print(d)
    for n, v in sorted(by.items()):
        v = np.array(v)
        print('  %-28s n=%4d  median %7.0f  p90 %7.0f  max %7.0f  mean %7.0f us  sum %.1f ms' % (n, len(v), np.median(v), np.percentile(v, 90), v.max(), v.mean(), v.sum() / 1e3))
