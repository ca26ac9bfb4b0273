# 16-row tiles x S = 4 (4-wave workgroups, bf16x3 arithmetic) against the default 32-row tiles x S = 8
run() { tag="$1"; shift; envs="$1"; shift; env $envs python tools/kernel_probe.py "$@" --tag "$tag [$envs]" 2>&1 | tail -1; }
for rep in 1 2 3; do
run h512 "X=1"
run h512 "CFL_DEBUG_PROJ_ROWS16=1 CFL_DEBUG_S=4"
run h512 "CFL_DEBUG_PROJ_ROWS16=1 CFL_DEBUG_S=8"
run h512 "CFL_DEBUG_S=4"
run hwn "X=1" --weight-norm
done
