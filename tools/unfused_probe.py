import os, sys, time
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'compatibility-family-learning_amd')]
import numpy as np, torch
from cfl import hipabi as H
from cfl.engine import PairEngine
from oracle import cfl_oracle as O
B,D,K,L=512,4096,3,20
cfg=O.EncoderCfg(D=D,L=L,K=K)
eng=PairEngine(D,L,K,norm=H.make_norm(1/58.388599),params=O.init_encoder_params(cfg,np.random.RandomState(0),np.float32),batch_size=B)
g=torch.Generator(device='cuda'); g.manual_seed(1)
nb=12
pool=[tuple(torch.randn(B,D,generator=g,device='cuda').abs_()*13 for _ in range(4)) for _ in range(nb)]
for mode in ('fused','unfused'):
    for i in range(50):
        if mode=='fused': eng.step(pool[i%nb])
        else: eng.fwd_bwd(pool[i%nb]); eng.apply_adam(1.0)
    torch.cuda.synchronize(); t0=time.perf_counter()
    n=1000
    for i in range(n):
        if mode=='fused': eng.step(pool[i%nb])
        else: eng.fwd_bwd(pool[i%nb]); eng.apply_adam(1.0)
    torch.cuda.synchronize(); print(mode, round((time.perf_counter()-t0)/n*1e6,2),'us/step')
