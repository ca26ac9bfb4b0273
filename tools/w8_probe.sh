#!/bin/bash
# eight-wave vs four-wave unsplit half tiles of the weight gradient (CFL_DEBUG_GRAD_W8=-1 disables), per-kernel probe
for v in 0 -1 0 -1; do
  export CFL_DEBUG_GRAD_W8=$v
  python tools/kernel_probe.py --tag "headline w8=$v" 2>&1 | tail -1
  python tools/kernel_probe.py --weight-norm --tag "headline-wn w8=$v" 2>&1 | tail -1
  python tools/kernel_probe.py --batch-size 1024 --tag "B1024 w8=$v" 2>&1 | tail -1
  python tools/kernel_probe.py --batch-size 100 --latent-size 10 --num-components 4 --tag "ref-shape w8=$v" 2>&1 | tail -1
  python tools/kernel_probe.py --latent-size 10 --num-components 4 --tag "L10K4 w8=$v" 2>&1 | tail -1
done
