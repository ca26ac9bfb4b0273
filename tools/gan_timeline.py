#!/usr/bin/env python
"""Timeline analysis of a rocprofv3 kernel trace of MrCGAN steps (tools/gan_trace.sh): per queue busy time, concurrency
histogram, idle gaps, and the biggest kernels of the last complete step.  Usage: python tools/gan_timeline.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    n = r['Kernel_Name']
    r['short'] = re.sub(r'\(.*', '', n.replace('void ', '')).replace('(anonymous namespace)::', '')[:70]
rows.sort(key=lambda r: r['s'])
# steps are delimited by the Adam launches of the generator (cfl_adam_kernel appears twice per step: D then G or G then D)
adam = [i for i, r in enumerate(rows) if 'cfl_adam' in r['Kernel_Name']]
print('launches %d, adam launches %d' % (len(rows), len(adam)))
# take the window between the 2nd-last pair and the last pair of adam launches = one full step
if len(adam) >= 4:
    lo = rows[adam[-3]]['e']; hi = rows[adam[-1]]['e']
else:
    lo = rows[0]['s']; hi = rows[-1]['e']
step = [r for r in rows if r['s'] >= lo and r['e'] <= hi + 1]
T = (hi - lo) / 1e6
print('step window %.3f ms, %d launches' % (T, len(step)))
ev = []
for r in step:
    ev.append((r['s'], 1)); ev.append((r['e'], -1))
ev.sort()
conc = collections.Counter(); cur = 0; last = lo
for t, d in ev:
    conc[cur] += t - last; last = t; cur += d
conc[cur] += hi - last
print('concurrency (kernels in flight): ' + '  '.join('%d: %.2f ms' % (k, v / 1e6) for k, v in sorted(conc.items())))
byq = collections.defaultdict(list)
for r in step:
    byq[(r['Queue_Id'], r['Stream_Id'])].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r['e'] - r['s'] for r in rs)
    print('queue %s stream %s: %4d launches, busy %.2f ms, first %.2f last %.2f' % (q[0], q[1], len(rs), busy / 1e6,
          (rs[0]['s'] - lo) / 1e6, (rs[-1]['e'] - lo) / 1e6))
agg = collections.Counter(); cnt = collections.Counter()
for r in step:
    agg[r['short']] += r['e'] - r['s']; cnt[r['short']] += 1
print('kernel time in the step: %.2f ms' % (sum(agg.values()) / 1e6))
for k, v in agg.most_common(25):
    print('  %-72s %4d  %8.1f us' % (k, cnt[k], v / 1e3))
if len(sys.argv) > 2:      # dump the step's launches in time order
    for r in step:
        print('%8.3f %8.3f q%s s%s %s grid %s' % ((r['s'] - lo) / 1e6, (r['e'] - r['s']) / 1e6, r['Queue_Id'], r['Stream_Id'], r['short'],
                                            r['Grid_Size_X']))
