for B in 1536 2048 3072; do
python tools/kernel_probe.py --batch-size $B --steps 200 --tag b${B} 2>&1 | tail -1
CFL_DEBUG_X3_KEEP_MB=240 CFL_DEBUG_PROJ_X3=1 python tools/kernel_probe.py --batch-size $B --steps 200 --tag b${B}_x3_keep240 2>&1 | tail -1
CFL_DEBUG_X3_KEEP_MB=240 CFL_DEBUG_X3_UNITS=256 CFL_DEBUG_PROJ_X3=1 python tools/kernel_probe.py --batch-size $B --steps 200 --tag b${B}_x3_keep240_u256 2>&1 | tail -1
done
CFL_DEBUG_X3_KEEP_MB=300 python tools/kernel_probe.py --batch-size 4096 --steps 100 --tag b4096_keep300 2>&1 | tail -1
python tools/kernel_probe.py --batch-size 4096 --steps 100 --tag b4096 2>&1 | tail -1
