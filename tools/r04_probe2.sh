# round 4, second probe: operand split of the bx3 projection (A truncated vs rounded), config-4 gradient tile shapes,
# CLI loop breakdown, MrCGAN profile
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
run() { tag="$1"; shift; envs="$1"; shift; env $envs python tools/kernel_probe.py "$@" --tag "$tag [$envs]" 2>&1 | tail -1; }
for rep in 1 2; do
bash tools/c34_probe.sh libcfl_hip.so libcfl_hip_arne.so
run c4 "CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=2" $C4
run c4 "CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=1" $C4
run c4 "CFL_DEBUG_P=2" $C4
run h512 "CFL_DEBUG_S=4"
run h512 "CFL_DEBUG_S=16"
done
