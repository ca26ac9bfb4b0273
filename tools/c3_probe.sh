C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100"
python tools/kernel_probe.py $C3 --weight-norm --tag c3 2>&1 | tail -1
CFL_DEBUG_NOFUSE=1 python tools/kernel_probe.py $C3 --weight-norm --tag c3_nofuse 2>&1 | tail -1
python tools/kernel_probe.py $C3 --tag c3_nown 2>&1 | tail -1
CFL_DEBUG_NOFUSE=1 python tools/kernel_probe.py $C3 --tag c3_nown_nofuse 2>&1 | tail -1
CFL_DEBUG_P=1 python tools/kernel_probe.py $C3 --weight-norm --tag c3_P1 2>&1 | tail -1
CFL_DEBUG_P=4 python tools/kernel_probe.py $C3 --weight-norm --tag c3_P4 2>&1 | tail -1
CFL_DEBUG_S=1 python tools/kernel_probe.py $C3 --weight-norm --tag c3_S1 2>&1 | tail -1
CFL_DEBUG_MID_NOROW=1 python tools/kernel_probe.py $C3 --weight-norm --tag c3_midnorow 2>&1 | tail -1
