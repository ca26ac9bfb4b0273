#!/bin/bash
# round 6, first measurement of the new tree: the new tests, the default bench line, the headline A/B against the round-5 library
set -u
mkdir -p gpurun_out/r06_a
python -m pytest tests/test_conv_gpu.py -m gpu -q -k "fcpcd or convpcd" 2>&1 | tail -3
python -m pytest tests/test_data_parallel_gpu.py -m gpu -q -k "bench_starts" 2>&1 | tail -15
python bench.py > gpurun_out/r06_a/bench.json 2> gpurun_out/r06_a/bench.err; echo "bench rc $?"; tail -c 1500 gpurun_out/r06_a/bench.err
LIBS="libcfl_hip.so libcfl_hip_nodp.so" bash tools/ab_lib.sh 2>&1 | tail -8
