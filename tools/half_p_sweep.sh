for B in 768 1024 1536 2048 3072; do
python tools/kernel_probe.py --batch-size $B --tag b${B}_default 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=2 python tools/kernel_probe.py --batch-size $B --tag b${B}_half_P2 2>&1 | tail -1
done
CFL_DEBUG_GRAD_HALF=1 CFL_DEBUG_P=4 python tools/kernel_probe.py --batch-size 3072 --tag b3072_half_P4 2>&1 | tail -1
CFL_DEBUG_GRAD_HALF=-1 python tools/kernel_probe.py --batch-size 3072 --tag b3072_nohalf 2>&1 | tail -1
