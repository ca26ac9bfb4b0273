#!/bin/bash
# round 6, fifth measurement: the child-failure test, how long the default bench run takes, the B sweep of the final kernels
set -u
R=$(pwd); O=$R/gpurun_out/r06_f; mkdir -p $O
timeout 900 python -m pytest tests/test_data_parallel_gpu.py -m gpu -q -k "dying or bench_starts" 2>&1 | tail -3
/usr/bin/time -v python bench.py > $O/bench.json 2> $O/bench.err; grep "Elapsed (wall clock)" $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['final_loss'], d['roofline']['traffic'], d['eval_auc']['hip'], d['other_configs']['config5_mrcgan_64x64_b100'].get('ms_per_step'))"
timeout 900 python tools/b_sweep.py > $O/b_sweep.md 2> $O/b_sweep.err; cat $O/b_sweep.md
