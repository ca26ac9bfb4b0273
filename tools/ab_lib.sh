for i in 1 2 3; do
for lib in ${LIBS:-libcfl_hip.so libcfl_hip_old.so}; do
CFL_HIP_LIB=/root/repo/compatibility-family-learning_amd/lib/$lib python bench.py --no-other-configs --no-cpu-baseline --no-cli-loop 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernels'])"
done; done
