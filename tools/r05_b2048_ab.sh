# round 5: unsplit 8-wave half tiles (P = 1) against the rows-split-in-two default beyond 2048 rows per side
run() { env $2 python bench.py --batch-size $3 --pool-mib 768 --timed-seconds 1.0 --no-other-configs --no-cpu-baseline --no-cli-loop --no-dp-form 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('B=%-5s %-28s %8.3f us/step  proj %.2f mid %.2f grad %.2f  %s' % ('$3', '$1', 1e3*d['ms_per_step'], k['proj']['avg_us'], k['mid']['avg_us'], k['grad']['avg_us'], d['roofline']['plan']['grad']))"; }
for B in ${BS:-1280 1536 2048 2560 3072}; do
for i in 1 2; do
run "default" "CFL_X=0" $B
run "P=1, 8 waves" "CFL_DEBUG_P=1 CFL_DEBUG_GRAD_HALF=1" $B
done; done
