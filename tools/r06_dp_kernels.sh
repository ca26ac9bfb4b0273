#!/bin/bash
# rocprofv3 kernel durations of the data-parallel step on a one-rank group, both exchanges
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06_h; mkdir -p $O
cd /tmp
for ex in allreduce oneshot; do
  CFL_DP_EXCHANGE=$ex python3 $R/tools/dp_probe.py 300 2>/dev/null | tail -1
  CFL_DP_EXCHANGE=$ex rocprofv3 --kernel-trace --stats --output-format csv -d $O/$ex -o run -- python3 $R/tools/dp_probe.py 300 > $O/$ex.log 2>&1
  find $O/$ex -name "*kernel_stats.csv" -exec cp {} $O/dp_${ex}_kernel_stats.csv \;
  rm -rf $O/$ex
  python3 - $O/dp_${ex}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print('  %-70s %5d calls %8.2f us avg' % (r['Name'][:70], int(r['Calls']), float(r['AverageNs']) / 1e3))
PY
done
