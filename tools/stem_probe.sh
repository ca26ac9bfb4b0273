# round 5: the image stem on its own kernels (csrc/conv_stem.h) -- per-layer table, step time with and without
CFL_GAN_OVERLAP=0 N=5 python tools/gan_layers_probe.py 2>&1 | grep -E "^step|Dis:conv/Conv "
for i in 1 2 3; do
  for e in CFL_DEBUG_NOSTEM=0 CFL_DEBUG_NOSTEM=1; do
    echo -n "$e  "; env $e N=20 python tools/gan_probe.py 2>&1 | grep -o "MrCGAN step B=100: [0-9.]* ms"
  done
done
