#!/bin/bash
# rocprofv3 kernel stats of the MrCGAN post-epoch step (tools/gan_probe.py) + launches per step (run on the GPU box).
# Usage: bash tools/gan_profile.sh <tag>
set -u
tag=$1
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
N=20 python3 tools/gan_probe.py > $O/gan.log 2>&1
cd /tmp
CFL_GAN_TUNE_STREAMS=0 N=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gan_stats -o run -- python3 $R/tools/gan_probe.py > $O/gan_prof.log 2>&1
cd $R
find $O/gan_stats -name "*kernel_stats.csv" -exec cp {} $O/mrcgan_kernel_stats.csv \;
find $O/gan_stats -name "*_kernel_trace.csv" -delete 2>/dev/null
python3 - $O/mrcgan_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = sum(int(r['Calls']) for r in rows)
steps = 2 + 20 + 20     # warm-up + timed + enqueue-only loops of tools/gan_probe.py (CFL_GAN_TUNE_STREAMS=0: no trial steps in the count)
print('kernel launches: %d in %d steps = %.0f per step' % (calls, steps, calls / steps))
PY
grep "MrCGAN step" $O/gan.log | cut -c1-60
