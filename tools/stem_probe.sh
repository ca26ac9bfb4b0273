# round 5: A/B of a MrCGAN step switch (same box, alternating): usage  bash tools/stem_probe.sh VAR=a VAR=b
for i in 1 2 3; do
  for e in "$@"; do
    echo -n "$e  "; env $e N=20 python tools/gan_probe.py 2>&1 | grep -o "MrCGAN step B=100: [0-9.]* ms"
  done
done
