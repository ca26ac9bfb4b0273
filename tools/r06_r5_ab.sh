#!/bin/bash
# round 6: same-box alternation of the round-5 tree (git worktree _r5tree at ebbce1b, its own library) and this tree:
# MrCGAN step (tools/gan_probe.py) and the headline step
set -u
for i in 1 2 3; do
  echo "r5   $(N=20 python _r5tree/tools/gan_probe.py 2>/dev/null | head -1 | cut -c1-40)"
  echo "r6   $(N=20 python tools/gan_probe.py 2>/dev/null | head -1 | cut -c1-40)"
  echo "r6 gp-late $(CFL_GAN_GP_EARLY=0 N=20 python tools/gan_probe.py 2>/dev/null | head -1 | cut -c1-40)"
done
for i in 1 2; do
  echo "r5 headline $(python _r5tree/bench.py --no-other-configs --no-cpu-baseline --no-cli-loop --no-dp-form --no-kernel-profile --timed-seconds 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])")"
  echo "r6 headline $(python bench.py --no-other-configs --no-cpu-baseline --no-cli-loop --no-dp-form --no-kernel-profile --timed-seconds 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])")"
done
