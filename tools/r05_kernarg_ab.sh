# round 5: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory instead of host-coherent memory) on the headline step
run() { env $1 python bench.py --timed-seconds 2.0 --no-other-configs --no-cpu-baseline --no-cli-loop --no-dp-form --no-kernel-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '%.3f us/step' % (1e3*d['ms_per_step']))"; }
for i in 1 2 3; do run HIP_FORCE_DEV_KERNARG=0; run HIP_FORCE_DEV_KERNARG=1; done
