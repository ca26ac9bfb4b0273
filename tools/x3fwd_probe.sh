# forward forms at training batch sizes: default vs forced bf16x3 forward (its W-plane launch is the `colnorm` entry),
# with x kept in cache (default when it fits the Infinity Cache) or streamed (CFL_DEBUG_PROJ_X3_KEEP=-1)
for B in 1024 1536 2048 3072; do
python tools/kernel_probe.py --batch-size $B --steps 200 --tag b${B} 2>&1 | tail -1
CFL_DEBUG_PROJ_X3=1 python tools/kernel_probe.py --batch-size $B --steps 200 --tag b${B}_x3fwd_keep 2>&1 | tail -1
CFL_DEBUG_PROJ_X3=1 CFL_DEBUG_PROJ_X3_KEEP=-1 python tools/kernel_probe.py --batch-size $B --steps 200 --tag b${B}_x3fwd_stream 2>&1 | tail -1
done
