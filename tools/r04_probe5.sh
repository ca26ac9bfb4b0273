# dL/dy pre-split by mid (plan.dy_pre): A/B per shape, one box
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
run() { tag="$1"; shift; envs="$1"; shift; env $envs python tools/kernel_probe.py "$@" --tag "$tag [$envs]" 2>&1 | tail -1; }
for rep in 1 2; do
for m in -1 1; do
run c3 "CFL_DEBUG_DY_PRE=$m" $C3
run c4 "CFL_DEBUG_DY_PRE=$m" $C4
run h512 "CFL_DEBUG_DY_PRE=$m"
run hwn "CFL_DEBUG_DY_PRE=$m" --weight-norm
run h1024 "CFL_DEBUG_DY_PRE=$m" --batch-size 1024
run c3pcd "CFL_DEBUG_DY_PRE=$m" --input-size 1024 --latent-size 64 --num-components 3 --weight-norm
done; done
