# config 3: row-split sweep of the paired half-tile weight gradient (per-kernel probe)
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
for i in 1 2; do
python tools/kernel_probe.py $C3 --tag "c3 default" 2>&1 | tail -1
CFL_DEBUG_P=2 CFL_DEBUG_GRAD_HALF=1 python tools/kernel_probe.py $C3 --tag "c3 half P=2" 2>&1 | tail -1
CFL_DEBUG_P=1 CFL_DEBUG_GRAD_HALF=1 python tools/kernel_probe.py $C3 --tag "c3 half P=1" 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=-1 python tools/kernel_probe.py $C3 --tag "c3 64-d tiles" 2>&1 | tail -1
done
