C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --caffe-margin 100 --weight-norm"
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
python tools/kernel_probe.py $C4 --tag c4 2>&1 | tail -1
CFL_DEBUG_PROJ_X3=1 python tools/kernel_probe.py $C4 --tag c4_x3fwd 2>&1 | tail -1
python tools/kernel_probe.py $C3 --tag c3 2>&1 | tail -1
CFL_DEBUG_PROJ_X3=1 python tools/kernel_probe.py $C3 --tag c3_x3fwd 2>&1 | tail -1
python tools/kernel_probe.py --tag hl 2>&1 | tail -1
CFL_DEBUG_PROJ_X3=1 python tools/kernel_probe.py --tag hl_x3fwd 2>&1 | tail -1
