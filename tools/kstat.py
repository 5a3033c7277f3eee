#!/usr/bin/env python3
"""print avg/min duration of kernels matching a substring from a rocprofv3 kernel_stats.csv tree:
   python tools/kstat.py <dir> <substring> [...]"""
import csv
import glob
import sys

for path in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if any(s in r['Name'] for s in sys.argv[2:]):
            print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us "
                  f"min {float(r['MinNs']) / 1e3:8.1f} us")
