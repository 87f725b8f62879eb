#!/usr/bin/env python3
"""Idle time between kernels in the steady state of a rocprofv3 kernel trace of bench.py (development tool).
usage: gap_analysis.py <kernel_trace.csv> [skip_first_markers] [marker-kernel-substring]  (marker: a kernel that runs once or a
fixed number of times per outer iteration; default the contraction passes)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
marker = sys.argv[3] if len(sys.argv) > 3 else 'contract'
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]) for r in rows), key=lambda e: e[0])
passes = [i for i, e in enumerate(ev) if marker in e[2]]
if len(passes) < skip + 4:
    sys.exit('too few passes')
lo, hi = passes[skip], passes[-4]
seg = ev[lo:hi + 1]
# idle = time covered by no kernel at all (kernels of the side stream overlap the passes)
t0, t1 = seg[0][0], seg[-1][0]
cur_end = seg[0][0]
idle = []
for st, en, name in seg:
    if st > cur_end:
        idle.append((st - cur_end, prev_name, name))
    if en > cur_end:
        cur_end, prev_name = en, name
npass = sum(1 for e in seg[:-1] if marker in e[2])
tot_idle = sum(g[0] for g in idle)
print('window: %d kernels, %d passes, span %.3f ms, no kernel running for %.3f ms = %.1f us per pass-to-pass interval'
      % (len(seg) - 1, npass, (t1 - t0) / 1e6, tot_idle / 1e6, tot_idle / 1e3 / max(npass, 1)))
for g in sorted(idle, key=lambda g: -g[0])[:14]:
    print('  idle %7.1f us after %-40s before %s' % (g[0] / 1e3, g[1][-40:], g[2][-40:]))
import collections
by = collections.defaultdict(list)
for g in idle:
    by[g[1].split('::')[-1][:36] + ' -> ' + g[2].split('::')[-1][:30]].append(g[0] / 1e3)
print('idle by kernel pair (us):')
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print('  %-70s n %4d mean %6.2f total %8.1f' % (k, len(v), sum(v) / len(v), sum(v)))
