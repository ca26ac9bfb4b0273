# forward forms at training batch sizes: default vs forced bf16x3 forward (its W-plane launch is the `colnorm` entry)
for B in 1024 2048 3072 4096; do
python tools/kernel_probe.py --batch-size $B --steps 200 --tag b${B} 2>&1 | tail -1
CFL_DEBUG_PROJ_X3=1 python tools/kernel_probe.py --batch-size $B --steps 200 --tag b${B}_x3fwd 2>&1 | tail -1
done
