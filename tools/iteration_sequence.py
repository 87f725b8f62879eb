#!/usr/bin/env python3
"""Kernel-by-kernel sequence of a few outer iterations from a rocprofv3 kernel trace: start (us), gap to the previous
kernel's end, duration, name -- between the <first>-th and the <last>-th contraction-pass launch.
usage: iteration_sequence.py <kernel_trace.csv> <first pass> <last pass>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r):
    return r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aoadmm::', '')[:40]
idx = [i for i, r in enumerate(rows) if nm(r).startswith('contract')]
a, b = idx[int(sys.argv[2])], idx[int(sys.argv[3])]
t0 = int(rows[a]['Start_Timestamp'])
prev = None
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f  +%6.1f  %7.2f  %s' % ((s - t0) / 1e3, (s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, nm(r)))
    prev = e
print('window: %.1f us for %d passes' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, int(sys.argv[3]) - int(sys.argv[2])))
