C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
run() { tag="$1"; shift; envs="$1"; shift; env $envs python tools/kernel_probe.py "$@" --tag "$tag [$envs]" 2>&1 | tail -1; }
for rep in 1 2; do
bash tools/c34_probe.sh libcfl_hip.so
run c4 "CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=4" $C4
run c3pcd "X=1" --input-size 1024 --latent-size 64 --num-components 3 --weight-norm
run h1024 "X=1" --batch-size 1024
done
