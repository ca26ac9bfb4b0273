# round 4: bf16x3 forward with planes kept by the Adam tail -- A/B per shape on one box
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
run() { tag="$1"; shift; envs="$1"; shift; env $envs python tools/kernel_probe.py "$@" --tag "$tag [$envs]" 2>&1 | tail -1; }
for rep in 1 2; do
run h512 "CFL_DEBUG_PROJ_BX3=-1"
run h512 "X=1"
run c3 "CFL_DEBUG_PROJ_BX3=-1" $C3
run c3 "X=1" $C3
run c4 "CFL_DEBUG_PROJ_BX3=-1" $C4
run c4 "X=1" $C4
run h1024 "CFL_DEBUG_PROJ_BX3=-1" --batch-size 1024
run h1024 "X=1" --batch-size 1024
run h2048 "CFL_DEBUG_PROJ_BX3=-1" --batch-size 2048
run h2048 "CFL_DEBUG_X3_KEPT_ROWS=99999" --batch-size 2048
run h256 "CFL_DEBUG_PROJ_BX3=-1" --batch-size 256
run h256 "X=1" --batch-size 256
run hwn "CFL_DEBUG_PROJ_BX3=-1" --weight-norm
run hwn "X=1" --weight-norm
run l10k4 "CFL_DEBUG_PROJ_BX3=-1" --latent-size 10 --num-components 4 --batch-size 100
run l10k4 "X=1" --latent-size 10 --num-components 4 --batch-size 100
done
