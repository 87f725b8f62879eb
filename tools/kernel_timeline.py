#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace: per-kernel totals per outer iteration and the non-tensor time.
usage: kernel_timeline.py <kernel_trace.csv> [n_iters]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
niter = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# steady state: from the last third of contract launches
idx = [i for i, r in enumerate(rows) if 'contract' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aoadmm::', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[n][0] += 1; agg[n][1] += d
tot = sum(v[1] for v in agg.values())
print('%-40s %8s %12s %10s' % ('kernel', 'calls', 'total_us', 'avg_us'))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-40s %8d %12.1f %10.2f' % (n[:40], c, t, t / c))
small = sum(t for n, (c, t) in agg.items() if not n.startswith('contract') and not n.startswith('synth') and 'tensor_sumsq' not in n)
print('non-tensor kernel time total: %.1f us  (per iteration if n_iters given: %.1f us)' % (small, small / niter))
