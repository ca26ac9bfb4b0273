# round 5: from how many rows per call should a training step with kept planes take the LDS-shared projection (cfl_proj_x3_keep_kernel)
# instead of the chunk-at-a-time one (cfl_proj_bx3_kernel)?  CFL_DEBUG_X3_KEPT_ROWS sweeps the threshold (default 3072).
run() { env $2 python bench.py --batch-size $3 --pool-mib 768 --timed-seconds 1.0 --no-other-configs --no-cpu-baseline --no-cli-loop --no-dp-form 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('B=%-5s %-22s %8.3f us/step  proj %.2f mid %.2f grad %.2f  %s S=%d' % ('$3', '$1', 1e3*d['ms_per_step'], k['proj']['avg_us'], k['mid']['avg_us'], k['grad']['avg_us'], d['roofline']['plan']['proj'], d['roofline']['plan']['S']))"; }
for B in ${BS:-768 1024 1280}; do
for i in 1 2; do
run "default" "CFL_X=0" $B
run "x3 from 1024 rows" "CFL_DEBUG_X3_KEPT_ROWS=1024" $B
done; done
