# round 5: the MrCGAN step's plan knobs again, now that the gradient-penalty chain runs beside the generator forward
run() { echo -n "$1  "; env $1 N=20 python tools/gan_probe.py 2>&1 | grep -o "MrCGAN step B=100: [0-9.]* ms"; }
for i in 1 2; do
run "CFL_X=0"
run "CFL_DEBUG_HALO_SPLIT_WGS=64"
run "CFL_DEBUG_HALO_SPLIT_WGS=256"
run "CFL_DEBUG_HALO_WGRAD_WGS=1024"
run "CFL_DEBUG_HALO_WGRAD_WGS=4096"
run "CFL_DEBUG_HALO_TN=128"
run "CFL_GAN_PREP_AHEAD=0"
done
