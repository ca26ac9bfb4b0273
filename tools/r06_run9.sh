# round 6: bench --gpus 2 on the ONE GPU of the box (two ranks share it, gloo): functional record of the dp_scaling legs incl. the
# sharded MrCGAN step; numbers are NOT scaling measurements (the ranks time-slice one GPU)
mkdir -p gpurun_out
CFL_DIST_BACKEND=gloo CFL_DP_MAX_BLOCKS=64 CFL_BENCH_LEG_SECONDS=0.5 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --repeats 5 --pool-mib 64 --no-cpu-baseline --no-kernel-profile --no-cli-loop > gpurun_out/r06_f_dp2_shared_gpu_bench.json 2> gpurun_out/r06_f_dp2.err
tail -c 3000 gpurun_out/r06_f_dp2_shared_gpu_bench.json
python - <<'PY'
import sys, numpy as np, torch
sys.path[:0]=['/root/repo','/root/repo/compatibility-family-learning_amd']
from cfl.models.gan_blocks import Generator, Discriminator
g=Generator('srgan',(64,64,3),84,'tanh',np.random.RandomState(0),torch.device('cuda'))
d=Discriminator('srgan',(64,64,3),64,np.random.RandomState(0),torch.device('cuda'))
print('config5 pool floats: G', g.pool.total, 'D', d.pool.total)
PY
