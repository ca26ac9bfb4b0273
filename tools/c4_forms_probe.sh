# config 4: forward form / slab count sweeps (per-kernel probe)
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
for i in 1 2; do
python tools/kernel_probe.py $C4 --tag "c4 default" 2>&1 | tail -1
CFL_DEBUG_X3_KEPT_ROWS=4096 python tools/kernel_probe.py $C4 --tag "c4 bx3 (kept rows 4096)" 2>&1 | tail -1
CFL_DEBUG_X3_KEPT_ROWS=4096 CFL_DEBUG_S=4 python tools/kernel_probe.py $C4 --tag "c4 bx3 S=4" 2>&1 | tail -1
CFL_DEBUG_X3_KEPT_ROWS=4096 CFL_DEBUG_S=2 python tools/kernel_probe.py $C4 --tag "c4 bx3 S=2" 2>&1 | tail -1
CFL_DEBUG_S=2 python tools/kernel_probe.py $C4 --tag "c4 x3 S=2" 2>&1 | tail -1
CFL_DEBUG_S=4 python tools/kernel_probe.py $C4 --tag "c4 x3 S=4" 2>&1 | tail -1
done
