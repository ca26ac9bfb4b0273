#!/usr/bin/env python
"""Summarise a rocprofv3 kernel trace of the pair step: per-kernel duration and the idle gap in front of each launch of a step
(proj -> mid -> grad), from the LAST 150 steps of the trace.  Usage: trace_gaps.py <..._kernel_trace.csv>"""
import csv
import sys
import statistics as st

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'cfl_' not in n:
        continue
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n.split('(')[0].split('<')[0]))
rows.sort()
rows = rows[-450:]
dur, gap = {}, {}
for i, (s, e, n) in enumerate(rows):
    dur.setdefault(n, []).append(e - s)
    if i:
        gap.setdefault(n, []).append(s - rows[i - 1][1])
step = []
names = [n for _, _, n in rows]
first = names[0]
starts = [s for s, _, n in rows if n == first]
for a, b in zip(starts, starts[1:]):
    step.append(b - a)
for n in dur:
    print('%-40s dur med %.2f us  (min %.2f)   gap-before med %.2f us (min %.2f)' % (
        n, st.median(dur[n]) / 1e3, min(dur[n]) / 1e3, st.median(gap.get(n, [0])) / 1e3, min(gap.get(n, [0])) / 1e3))
if step:
    print('step period med %.2f us (min %.2f), %d steps' % (st.median(step) / 1e3, min(step) / 1e3, len(step)))
