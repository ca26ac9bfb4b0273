C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
python tools/kernel_probe.py $C3 --tag c3_default 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=-1 python tools/kernel_probe.py $C3 --tag c3_nohalf 2>&1 | tail -1
python tools/kernel_probe.py $C4 --tag c4_default 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=2 python tools/kernel_probe.py $C4 --tag c4_half_P2 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=1 python tools/kernel_probe.py $C4 --tag c4_half_P1 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=4 python tools/kernel_probe.py $C4 --tag c4_half_P4 2>&1 | tail -1
python tools/kernel_probe.py --batch-size 4096 --tag b4096_default 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=2 python tools/kernel_probe.py --batch-size 4096 --tag b4096_half_P2 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=4 python tools/kernel_probe.py --batch-size 4096 --tag b4096_half_P4 2>&1 | tail -1
python tools/kernel_probe.py --batch-size 2048 --tag b2048_default 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=2 python tools/kernel_probe.py --batch-size 2048 --tag b2048_half_P2 2>&1 | tail -1
