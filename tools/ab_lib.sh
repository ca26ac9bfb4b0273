# same-box alternation of library variants on the headline step (tools/build_variant.sh builds them): LIBS="libcfl_hip.so libcfl_hip_<name>.so"
# (--no-dp-form: an ablation build need not be consistent with the data-parallel legs)
for i in 1 2 3; do
for lib in ${LIBS:-libcfl_hip.so libcfl_hip_nodp.so}; do
CFL_HIP_LIB=/root/repo/compatibility-family-learning_amd/lib/$lib timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-cli-loop --no-dp-form --no-live-traffic --timed-seconds 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernels'])"
done; done
