#!/bin/bash
# round 6, second measurement: (1) grid barrier vs kernel boundary microbenchmark, (2) the x-split ablation A/B at B = 512 / 2048 / 8192,
# (3) config 0's conv model through the CLI under rocprofv3: kernel time per iteration beside the wall time
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06_b; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/microbench/gridbar.hip -o /tmp/gridbar && /tmp/gridbar > $O/gridbar.txt 2>&1
cat $O/gridbar.txt
bash tools/r06_xsplit_ab.sh > $O/xsplit_ab.txt 2>&1
cat $O/xsplit_ab.txt
LIBS="libcfl_hip.so libcfl_hip_nodp.so" bash tools/ab_lib.sh > $O/nodp_ab.txt 2>&1
cat $O/nodp_ab.txt
timeout 900 python -m pytest tests/test_data_parallel_gpu.py -m gpu -q -k "bench_starts" 2>&1 | tail -5
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-cli-loop --no-kernel-profile --timed-seconds 2 > $O/bench_dp_form.json 2> $O/bench_dp_form.err; python -c "import json; d=json.load(open('$O/bench_dp_form.json')); print(d['ms_per_step'], d['config']['final_loss'], d['config']['loss_at_restore']); print(json.dumps(d['dp_form'])[:1500])"

timeout 600 python3 tools/conv_epoch_probe.py 20000 2>/dev/null | tail -1 > $O/conv_epoch_wall.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/conv -o run -- python3 $R/tools/conv_epoch_probe.py 20000 > $O/conv_prof.log 2>&1
find $O/conv -name "*kernel_stats.csv" -exec cp {} $O/conv_epoch_kernel_stats.csv \;
rm -rf $O/conv
cd $R
cat $O/conv_epoch_wall.txt; tail -2 $O/conv_prof.log
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r06_b/conv_epoch_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
iters = 4 * 200          # the probe trains 1 + 3 epochs of 200 iterations
print('kernel time %.3f ms per iteration, %d launches per iteration (all kernels of the run / %d iterations)' % (tot / iters / 1e6, calls // iters, iters))
for r in rows[:12]:
    print('%-90s %6d calls  %8.1f us avg  %5.1f %%' % (r['Name'][:90], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
