#!/bin/bash
# rocprofv3 kernel TRACE (start / end / queue of every launch) of a few MrCGAN steps: the timeline behind tools/gan_timeline.py
# Usage: bash tools/gan_trace.sh <tag>     (on the GPU box)
set -u
tag=$1
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp
N=3 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 $R/tools/gan_probe.py > $O/gan_trace.log 2>&1
cd $R
find $O/trace -name "*_kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
ls -la $O
