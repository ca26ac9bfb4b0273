#!/bin/bash
# Run on the GPU box: the scoring path (one dist_eval-sized call = proj + mid over RESIDENT_EVAL_ROWS pairs):
# event-timed call, rocprofv3 kernel stats, two PMC passes (HBM fetch / write).  Usage: bash tools/measure_eval.sh <tag> [pairs]
set -u
tag=${1:-e}
pairs=${2:-32768}
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
python3 tools/score_loop.py --pairs $pairs --calls 200 > $O/${tag}_eval_call.json 2> $O/${tag}_eval_call.err
python3 tools/score_loop.py --pairs $pairs --calls 200 --indexed > $O/${tag}_eval_call_indexed.json 2>> $O/${tag}_eval_call.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_eval_stats -o run -- python3 $R/tools/score_loop.py --pairs $pairs --calls 200 > $O/${tag}_eval_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_eval_fetch -o run -- python3 $R/tools/score_loop.py --pairs $pairs --calls 60 > $O/${tag}_eval_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_eval_write -o run -- python3 $R/tools/score_loop.py --pairs $pairs --calls 60 > $O/${tag}_eval_write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/${tag}_eval_fetch $O/${tag}_eval_write $O/${tag}_traffic_eval.json > $O/${tag}_traffic_eval_detail.json
find $O/${tag}_eval_stats -name "*kernel_stats.csv" -exec cp {} $O/${tag}_eval_kernel_stats.csv \;
find $O/${tag}_eval_stats $O/${tag}_eval_fetch $O/${tag}_eval_write -name "*_kernel_trace.csv" -delete 2>/dev/null
find $O/${tag}_eval_fetch $O/${tag}_eval_write -name "*counter_collection.csv" -delete 2>/dev/null
cat $O/${tag}_eval_call.json $O/${tag}_eval_call_indexed.json
head -6 $O/${tag}_eval_kernel_stats.csv
cat $O/${tag}_traffic_eval.json
