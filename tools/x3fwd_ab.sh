for B in 1536 2048 3072 4096; do
echo "B=$B"; bash tools/ab_env.sh CFL_DEBUG_PROJ_X3=-1 --batch-size $B --pool-mib $((B*4096*16*3/1048576)) 2>&1 | tail -4
done
echo "B=4096 keep off"; bash tools/ab_env.sh CFL_DEBUG_PROJ_X3_KEEP=-1 --batch-size 4096 --pool-mib 800 2>&1 | tail -4
for P in 2048 4096; do
python tools/score_loop.py --pairs $P --calls 200 | tail -1
CFL_DEBUG_PROJ_X3=1 python tools/score_loop.py --pairs $P --calls 200 | tail -1
done
