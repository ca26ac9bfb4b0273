#!/bin/bash
# round 6: per-kernel durations AND the gaps between the launches of a step at the reference's own shape (B = 100, K = 4, L = 10)
# and at the headline shape: rocprofv3 kernel trace of tools/kernel_probe.py, summarised by tools/trace_gaps.py
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06_trace; mkdir -p $O
cd /tmp
for cfg in "b100 --batch-size 100 --num-components 4 --latent-size 10" "b512 --batch-size 512 --num-components 3 --latent-size 20"; do
  set -- $cfg; tag=$1; shift
  python3 $R/tools/kernel_probe.py "$@" --steps 300 --tag $tag 2>&1 | tail -1 > $O/$tag.probe.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o run -- python3 $R/tools/kernel_probe.py "$@" --steps 200 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*_kernel_trace.csv" | head -1)
  python3 $R/tools/trace_gaps.py "$f" > $O/$tag.gaps.txt 2>&1
  find $O/$tag -name "*kernel_stats.csv" -exec cp {} $O/$tag.kernel_stats.csv \;
  rm -rf $O/$tag
done
cd $R
cat $O/*.probe.json $O/*.gaps.txt
