C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
run() { env $2 python tools/kernel_probe.py $3 --steps 400 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %-20s step %.2f us  %s' % ('$1', '$2', d['step_us'], d['kernels_us']))"; }
for i in 1 2 3; do
run c4 "CFL_X=0" "$C4"; run c4 "CFL_DEBUG_S=2" "$C4"
done
