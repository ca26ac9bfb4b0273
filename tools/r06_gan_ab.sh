#!/bin/bash
# MrCGAN step, same box, stream-placement tuner OFF (deterministic placement per process): round-5 tree / round 6 with the deferred
# finalisation (in groups of CFL_GAN_DEFER_GROUP layers) and the batched cache preparation on or off
set -u
export CFL_GAN_TUNE_STREAMS=0
for i in 1 2 3; do
  echo "r5 tree            $(N=30 python _r5tree/tools/gan_probe.py 2>/dev/null | head -1 | cut -c19-26)"
  for grp in 3 6 12 64; do
  echo "r6 both, group $grp   $(CFL_GAN_DEFER_GROUP=$grp N=30 python tools/gan_probe.py 2>/dev/null | head -1 | cut -c19-26)"
  done
  echo "r6 prep only       $(CFL_GAN_DEFER_WFINAL=0 N=30 python tools/gan_probe.py 2>/dev/null | head -1 | cut -c19-26)"
  echo "r6 neither         $(CFL_GAN_DEFER_WFINAL=0 CFL_GAN_PREP_BATCHED=0 N=30 python tools/gan_probe.py 2>/dev/null | head -1 | cut -c19-26)"
done
