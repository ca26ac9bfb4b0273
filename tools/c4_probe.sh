C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
python tools/kernel_probe.py $C4 --tag c4 2>&1 | tail -1
CFL_DEBUG_S=4 python tools/kernel_probe.py $C4 --tag c4_S4 2>&1 | tail -1
CFL_DEBUG_S=1 python tools/kernel_probe.py $C4 --tag c4_S1 2>&1 | tail -1
CFL_DEBUG_P=8 python tools/kernel_probe.py $C4 --tag c4_P8 2>&1 | tail -1
CFL_DEBUG_P=2 python tools/kernel_probe.py $C4 --tag c4_P2 2>&1 | tail -1
CFL_DEBUG_PROJ_STREAM=1 python tools/kernel_probe.py $C4 --tag c4_stream 2>&1 | tail -1
CFL_DEBUG_PROJ_X3=1 python tools/kernel_probe.py $C4 --tag c4_x3fwd 2>&1 | tail -1
CFL_DEBUG_S=4 CFL_DEBUG_P=8 python tools/kernel_probe.py $C4 --tag c4_S4P8 2>&1 | tail -1
