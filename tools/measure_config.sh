#!/bin/bash
# rocprofv3 kernel stats + PMC passes of tools/kernel_probe.py on one configuration (run on the GPU box).
# Usage: bash tools/measure_config.sh <tag> <kernel_probe args...>     (outputs under gpurun_out/<tag>/)
set -u
tag=$1; shift
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
python3 tools/kernel_probe.py "$@" --tag "$tag" 2>&1 | tail -1 > $O/probe.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/tools/kernel_probe.py "$@" --steps 300 > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/stats -name "*_kernel_trace.csv" -delete 2>/dev/null
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "TCP_PENDING_STALL_CYCLES TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -o run -- python3 $R/tools/kernel_probe.py "$@" --steps 40 > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY' >> $O/pmc_counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:60]
    if 'cfl_' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in acc:
    for c in sorted(acc[k]):
        print('%-62s %-36s per-dispatch %.1f  (n=%d)' % (k, c, acc[k][c] / cnt[(k, c)], cnt[(k, c)]))
PY
    rm -f "$f"
  else
    tail -3 $O/pmc$i.log >> $O/pmc_counters.txt
  fi
done
cd $R
cat $O/probe.json; grep cfl_ $O/kernel_stats.csv | head -6; cat $O/pmc_counters.txt
