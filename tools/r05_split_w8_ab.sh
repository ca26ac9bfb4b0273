# (CFL_DEBUG_SPLIT_W8 and cfl_grad_x3_half_split_w8_kernel were removed after this measurement: it loses; see LEDGER.md round 5, item 11)
# round 5: eight waves per workgroup for the hand-off forms of the half-tile weight gradient (configs 3 / 4)
C3="--input-size 1024 --latent-size 256 --num-components 1 --dist-type siamese --weight-norm --caffe-margin 100"
C4="--input-size 2048 --latent-size 20 --num-components 5 --weight-norm --batch-size 1024"
run() { env $2 python tools/kernel_probe.py $3 --steps 400 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %-12s step %.2f us  %s' % ('$1', '$2', d['step_us'], d['kernels_us']))"; }
for i in 1 2 3; do
run c3 "CFL_X=0" "$C3"; run c3 "CFL_DEBUG_SPLIT_W8=1" "$C3"
run c4 "CFL_X=0" "$C4"; run c4 "CFL_DEBUG_SPLIT_W8=1" "$C4"
done
