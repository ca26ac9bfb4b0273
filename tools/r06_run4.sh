#!/bin/bash
# round 6, fourth measurement: the merged Adam + gather kernel of the one-shot exchange (DP tests, one-rank forms), and the bench legs
set -u
R=$(pwd); O=$R/gpurun_out/r06_e; mkdir -p $O
timeout 1200 python -m pytest tests/test_dp_step_gpu.py tests/test_data_parallel_gpu.py -m gpu -q -x 2>&1 | tail -5
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-cli-loop --no-kernel-profile --timed-seconds 2 > $O/bench_dp_form.json 2> $O/bench_dp_form.err; python -c "import json; d=json.load(open('$O/bench_dp_form.json')); print(d['ms_per_step']); print(json.dumps(d['dp_form'])[:1700])"
CFL_DP_SPLIT_ADAM=1 timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-cli-loop --no-kernel-profile --timed-seconds 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split adam/gather:', json.dumps(d['dp_form']['oneshot'])[:300])"
