# configs 3 / 4 and the weight-normalised headline through the per-kernel probe, for each library given
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
for lib in "$@"; do
  export CFL_HIP_LIB=$PWD/compatibility-family-learning_amd/lib/$lib
  python tools/kernel_probe.py $C3 --tag "c3 $lib" 2>&1 | tail -1
  python tools/kernel_probe.py $C4 --tag "c4 $lib" 2>&1 | tail -1
  python tools/kernel_probe.py --weight-norm --tag "headline-wn $lib" 2>&1 | tail -1
  python tools/kernel_probe.py --tag "headline $lib" 2>&1 | tail -1
done
