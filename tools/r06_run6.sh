timeout 900 python -m pytest tests/test_dp_step_gpu.py tests/test_data_parallel_gpu.py -m gpu -q -x 2>&1 | tail -3
bash tools/r06_dp_kernels.sh 2>&1 | tail -20
