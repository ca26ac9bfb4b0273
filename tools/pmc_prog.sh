#!/bin/bash
# rocprofv3 kernel stats + PMC passes of any python program of tools/ (run on the GPU box).
# Usage: bash tools/pmc_prog.sh <tag> <prog.py> "<counters pass 1>" "<counters pass 2>" ...     (outputs under gpurun_out/<tag>/)
set -u
tag=$1; prog=$2; shift 2
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/$prog > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name "*_kernel_trace.csv" -delete 2>/dev/null
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -o run -- python3 $R/$prog > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY' >> $O/pmc_counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:70] + ' grid=' + r.get('Grid_Size', '?')
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in acc:
    for c in sorted(acc[k]):
        print('%-90s %-36s per-dispatch %.1f  (n=%d)' % (k, c, acc[k][c] / cnt[(k, c)], cnt[(k, c)]))
PY
    rm -f "$f"
  else
    tail -3 $O/pmc$i.log >> $O/pmc_counters.txt
  fi
done
cd $R
head -12 $O/kernel_stats.csv | cut -c1-200; cat $O/pmc_counters.txt
