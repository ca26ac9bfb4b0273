#!/bin/bash
# eight-wave vs four-wave half tiles with a hand-off (configs 3 / 4, B = 2048), per-kernel probe
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
for v in 0 -1 0 -1; do
  export CFL_DEBUG_GRAD_W8=$v
  python tools/kernel_probe.py $C3 --tag "c3 w8=$v" 2>&1 | tail -1
  python tools/kernel_probe.py $C4 --tag "c4 w8=$v" 2>&1 | tail -1
  python tools/kernel_probe.py --batch-size 2048 --tag "B2048 w8=$v" 2>&1 | tail -1
  python tools/kernel_probe.py --batch-size 1536 --tag "B1536 w8=$v" 2>&1 | tail -1
done
