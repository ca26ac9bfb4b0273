for i in 1 2; do
for v in default 1 0; do
if [ $v = default ]; then pre=""; else pre="HIP_FORCE_DEV_KERNARG=$v"; fi
env $pre python bench.py --no-other-configs --no-cpu-baseline --no-cli-loop 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernarg $v', d['ms_per_step'], d['roofline']['kernels'])"
done; done
