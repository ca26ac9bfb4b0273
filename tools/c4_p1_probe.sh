C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
for i in 1 2 3; do
python tools/kernel_probe.py $C4 --tag "c4 default" 2>&1 | tail -1
CFL_DEBUG_P=1 CFL_DEBUG_GRAD_HALF=1 python tools/kernel_probe.py $C4 --tag "c4 P=1 half w8" 2>&1 | tail -1
CFL_DEBUG_P=1 CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_GRAD_W8=-1 python tools/kernel_probe.py $C4 --tag "c4 P=1 half w4" 2>&1 | tail -1
done
