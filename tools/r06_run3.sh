#!/bin/bash
# round 6, third measurement: DP kernels without system fences (one-rank forms + launch counts), headline A/B against the library
# without any DP hook, the DP / GAN tests, conv epoch wall + kernel time, grid barrier microbenchmark (relaxed arrivals)
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06_c; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/microbench/gridbar.hip -o /tmp/gridbar && /tmp/gridbar > $O/gridbar.txt 2>&1; cat $O/gridbar.txt
timeout 900 python -m pytest tests/test_dp_step_gpu.py tests/test_data_parallel_gpu.py -m gpu -q -x 2>&1 | tail -5
LIBS="libcfl_hip.so libcfl_hip_nodp.so" bash tools/ab_lib.sh > $O/nodp_ab.txt 2>&1; cat $O/nodp_ab.txt
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-cli-loop --no-kernel-profile --timed-seconds 2 > $O/bench_dp_form.json 2> $O/bench_dp_form.err; python -c "import json; d=json.load(open('$O/bench_dp_form.json')); print(d['ms_per_step'], d['config']['final_loss'], d['config']['loss_at_restore']); print(json.dumps(d['dp_form'])[:1700])"
timeout 600 python3 tools/conv_epoch_probe.py 20000 2>/dev/null | tail -1 > $O/conv_epoch_wall.txt; cat $O/conv_epoch_wall.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/conv -o run -- python3 $R/tools/conv_epoch_probe.py 20000 > $O/conv_prof.log 2>&1
find $O/conv -name "*kernel_stats.csv" -exec cp {} $O/conv_epoch_kernel_stats.csv \;
rm -rf $O/conv
cd $R
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r06_c/conv_epoch_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
iters = 3 * 200
print('kernel time %.3f ms per iteration, %d launches per iteration (all kernels of the 3-epoch run / %d iterations)' % (tot / iters / 1e6, calls // iters, iters))
PY
