for B in 8192; do
python tools/kernel_probe.py --batch-size $B --steps 60 --tag b${B} 2>&1 | tail -1
CFL_DEBUG_X3_KEEP_MB=2000 python tools/kernel_probe.py --batch-size $B --steps 60 --tag b${B}_keep 2>&1 | tail -1
done
for B in 1280 1536 2048 3072; do
CFL_DEBUG_X3_KEEP_MB=300 CFL_DEBUG_PROJ_X3=1 python tools/kernel_probe.py --batch-size $B --steps 200 --tag b${B}_x3_u384 2>&1 | tail -1
done
python tools/kernel_probe.py --batch-size 1280 --steps 200 --tag b1280 2>&1 | tail -1
python tools/score_loop.py --pairs 32768 --calls 100 | tail -1
python tools/score_loop.py --pairs 8192 --calls 200 | tail -1
python tools/score_loop.py --pairs 4096 --calls 200 | tail -1
