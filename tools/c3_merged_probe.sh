# config 3: merged two-side gradient workgroups vs the hand-off form (CFL_DEBUG_GRAD_MERGED=-1)
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
for i in 1 2 3; do for v in 0 -1; do
CFL_DEBUG_GRAD_MERGED=$v python tools/kernel_probe.py $C3 --tag "c3 merged=$v" 2>&1 | tail -1
done; done
for v in 0 -1; do
CFL_DEBUG_GRAD_MERGED=$v python tools/kernel_probe.py --input-size 4096 --latent-size 64 --num-components 1 --dist-type siamese --tag "siamese D4096 L64 merged=$v" 2>&1 | tail -1
CFL_DEBUG_GRAD_MERGED=$v python tools/kernel_probe.py --input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --batch-size 1024 --tag "siamese B1024 merged=$v" 2>&1 | tail -1
done
