#!/bin/bash
# round 6: what could resident bf16 planes of the feature table save AT MOST?  Same-box alternation of the
# production library and the ABL_X_NOSPLIT build (x operands "arrive split": the split arithmetic of every x fragment removed from
# both contraction kernels, bytes unchanged, numbers wrong) at B = 512 / 2048 / 8192: step time and the event-timed kernel intervals.
set -u
L=/root/repo/compatibility-family-learning_amd/lib
for i in 1 2 3; do
for B in 512 2048 8192; do
for lib in libcfl_hip.so libcfl_hip_xnosplit.so; do
mib=$(( 3 * 4 * B * 4096 * 4 / 1048576 )); [ $mib -lt 384 ] && mib=384
CFL_HIP_LIB=$L/$lib timeout 300 python bench.py --batch-size $B --steps 20 --warmup 5 --pool-mib $mib --timed-seconds 1.5 --no-other-configs --no-cpu-baseline --no-cli-loop --no-dp-form --no-live-traffic 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$B', '$lib', round(1e3*d['ms_per_step'],2), {k:v['avg_us'] for k,v in d['roofline']['kernels'].items()})"
done; done; done
