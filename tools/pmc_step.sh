#!/bin/bash
# PMC passes over the headline training step (bench.py) (run on the GPU box).  Usage: bash tools/pmc_step.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -u
tag=$1; shift
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -o run -- python3 $R/bench.py --steps 60 --warmup 10 --repeats 1 --no-cpu-baseline --no-kernel-profile --no-cli-loop --no-other-configs --no-dp-form > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY' > $O/pmc$i.summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in acc:
    for c in sorted(acc[k]):
        print('%-62s %-36s per-dispatch %.1f  (n=%d)' % (k, c, acc[k][c] / cnt[(k, c)], cnt[(k, c)]))
PY
    cat $O/pmc$i.summary.txt | grep -i "cfl_"
    rm -f "$f"
  else
    tail -5 $O/pmc$i.log
  fi
done
