"""Aggregate rocprofv3 --pmc output (csv, one row per dispatch and counter) per kernel.

    python tools/pmc_summary.py <dir with *_counter_collection.csv> [more dirs ...] > profiles/rNN_pmc_<what>.csv

Prints one line per (kernel, counter): dispatches, sum, mean per dispatch.  Counters are summed over the
dimensions rocprofv3 reports (XCDs, SEs ...).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: [0, 0.0])
    disp = defaultdict(set)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(f, newline='') as fh:
                for row in csv.DictReader(fh):
                    k = row['Kernel_Name'].split('(')[0]
                    c = row['Counter_Name']
                    acc[(k, c)][1] += float(row['Counter_Value'])
                    disp[(k, c)].add((f, row['Dispatch_Id']))
    print('kernel,counter,dispatches,sum,mean_per_dispatch')
    for (k, c) in sorted(acc):
        n = len(disp[(k, c)])
        if k.startswith('k_'):
            print('%s,%s,%d,%.6g,%.6g' % (k, c, n, acc[(k, c)][1], acc[(k, c)][1] / max(n, 1)))


if __name__ == '__main__':
    main()
