#!/bin/bash
# Run on the GPU box: bench line + rocprofv3 kernel stats + two PMC passes (HBM fetch / write).
# Usage: bash tools/measure.sh <tag>     (outputs under gpurun_out/<tag>_*)
set -u
tag=${1:-m}
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/${tag}_bench.json 2> $O/${tag}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o run -- python3 $R/bench.py --steps 500 --warmup 50 --repeats 2 --no-cpu-baseline --no-kernel-profile --no-cli-loop --no-other-configs --no-dp-form --no-live-traffic > $O/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -o run -- python3 $R/bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-kernel-profile --no-cli-loop --no-other-configs --no-dp-form --no-live-traffic > $O/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -o run -- python3 $R/bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-kernel-profile --no-cli-loop --no-other-configs --no-dp-form --no-live-traffic > $O/${tag}_write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/${tag}_fetch $O/${tag}_write $O/${tag}_traffic.json > $O/${tag}_traffic_detail.json
find $O/${tag}_stats -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats.csv \;
# keep only small summaries (the traces are large)
find $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write -name "*_kernel_trace.csv" -delete 2>/dev/null
find $O/${tag}_fetch $O/${tag}_write -name "*counter_collection.csv" -delete 2>/dev/null
cat $O/${tag}_bench.json
head -8 $O/${tag}_kernel_stats.csv
cat $O/${tag}_traffic.json
