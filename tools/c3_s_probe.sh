# config 3: d-slice count sweep (per-kernel probe)
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
for i in 1 2; do
for s in 0 8 4 2; do
CFL_DEBUG_S=$s python tools/kernel_probe.py $C3 --tag "c3 S=$s" 2>&1 | tail -1
done; done
