#!/bin/bash
set -u
timeout 1500 python -m pytest tests/test_gan_gpu.py tests/test_activation_masks_gpu.py tests/test_cli_gpu.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2 3; do
  echo "batched    $(N=20 python tools/gan_probe.py 2>/dev/null | head -1 | cut -c1-40)"
  echo "per-layer  $(CFL_GAN_DEFER_WFINAL=0 CFL_GAN_PREP_BATCHED=0 N=20 python tools/gan_probe.py 2>/dev/null | head -1 | cut -c1-40)"
  echo "r5 tree    $(N=20 python _r5tree/tools/gan_probe.py 2>/dev/null | head -1 | cut -c1-40)"
done
bash tools/gan_profile.sh r06_gan 2>&1 | tail -3
