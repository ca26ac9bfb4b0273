# round 5: config 4 (192 half tiles < 256 CUs): rows split in two (384 workgroups x 4 waves, default) against unsplit eight-wave tiles (192 x 8 waves)
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
run() { env $2 python tools/kernel_probe.py $3 --steps 400 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %-44s step %.2f us  %s' % ('$1', '$2', d['step_us'], d['kernels_us']))"; }
for i in 1 2 3; do
run c4 "CFL_X=0" "$C4"; run c4 "CFL_DEBUG_P=1 CFL_DEBUG_GRAD_HALF=1" "$C4"
done
