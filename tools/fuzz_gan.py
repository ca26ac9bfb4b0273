#!/usr/bin/env python
"""MrCGAN blocks / post-epoch step on unusual image shapes against the torch-fp64 oracle (the assertions of
tests/test_gan_gpu.py).  A failure that disappears with another data seed is an lrelu mask flip: a pre-activation
within fp32 rounding of zero takes slope 1 in one precision and 0.2 in the other (seen for ('srgan', (32, 32, 2)),
seed 0, 5 other seeds clean)."""
import os, sys
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'compatibility-family-learning_amd')]
import tests.test_gan_gpu as T
fails=0
for gt, shape, me, mp, gp in [('srgan',(12,12,1),0.05,0.2,0.5), ('conv',(12,12,3),None,0.3,0.5), ('srgan',(24,24,3),None,None,None),
                              ('conv',(28,28,1),0.1,0.5,0.25), ('srgan',(32,32,2),0.05,None,1.0), ('conv',(8,8,4),None,None,0.5)]:
    try:
        T.test_post_epoch_step_matches_oracle(gt, shape, me, mp, gp)
        T.test_generator_fwd_bwd(gt, shape)
        T.test_discriminator_fwd_bwd_gp(gt, shape)
        print('ok', gt, shape)
    except Exception as e:
        fails+=1; print('FAIL', gt, shape, repr(e)[:300])
print('fails', fails)
