# round 5: d split of the chunk-at-a-time projection at intermediate batch sizes (default: 512 / (row tiles x jobs), a power of two)
run() { env $2 python bench.py --batch-size $3 --pool-mib 768 --timed-seconds 1.0 --no-other-configs --no-cpu-baseline --no-cli-loop --no-dp-form 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('B=%-5s %-16s %8.3f us/step  proj %.2f mid %.2f grad %.2f  S=%d xcd=%d' % ('$3', '$1', 1e3*d['ms_per_step'], k['proj']['avg_us'], k['mid']['avg_us'], k['grad']['avg_us'], d['roofline']['plan']['S'], d['roofline']['plan']['xcd_aligned']))"; }
for B in ${BS:-640 768 1024}; do
for i in 1 2; do
run "default" "CFL_X=0" $B
run "S=8" "CFL_DEBUG_S=8" $B
done; done
