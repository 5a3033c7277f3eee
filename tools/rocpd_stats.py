#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (the default output format of
`rocprofv3 --kernel-trace`): python tools/rocpd_stats.py <results.db> [name-filter]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    rows = con.execute('select name, count(*), avg(end-start), min(end-start), max(end-start), '
                       'sum(end-start) from kernels group by name order by sum(end-start) desc')
    print('Kernel,Calls,AverageNs,MinNs,MaxNs,TotalDurationNs')
    for name, n, avg, mn, mx, tot in rows:
        if flt and flt not in name:
            continue
        short = name.replace('void ', '').replace('nmsa::', '').replace('(anonymous namespace)::', '')
        print(f'"{short[:90]}",{n},{avg:.1f},{mn},{mx},{tot}')


if __name__ == '__main__':
    main()
