import sys, numpy as np, torch
sys.path[:0]=['/root/repo','/root/repo/compatibility-family-learning_amd']
from oracle import cfl_oracle as O, conv_oracle as CO
import tests.test_oracle as TO
from cfl import ops, hipabi as H
from cfl.models.cfl import construct_model
rng = np.random.RandomState(4)
B, shape, L, K, reg = 12, (28, 28, 1), 30, 1, 5e-4
dn = ops.dist_normalizer(shape, None, None, None, None, None, 'sigmoid')
kw = dict(is_double=False, disable_double=False, latent_shape=None, source_shape=None, input_shape=shape, ae_shape=None, batch_size=B, data_norm=None, data_type='sigmoid', model_type='conv', gan_type='conv', num_components=K, latent_size=L, pos_weight=None, caffe_margin=None, gan=False, cgan=False, t_dim=None, dist_type='pcd', act_type=None, use_threshold=True, lr=1e-3, beta1=0.9, beta2=0.999, z_dim=20, z_stddev=1., g_dim=64, g_lr=2e-4, g_beta1=.5, g_beta2=.999, m_prj=None, m_enc=None, d_dim=64, d_lr=2e-4, d_beta1=.5, d_beta2=.999, lambda_dra=.5, lambda_gp=None, lambda_m=0.0, directed=False, data_directed=False, reg_const=reg, data_normalizer=dn[0], data_unnormalizer=dn[1], seed=2)
model, _ = construct_model(**kw)
model.engine.theta[model.engine.layout.thr] = 0.3
hp, _, thr = model.engine.named_variables()
cfg = O.EncoderCfg(D=6272, L=L, K=K, dist_type='pcd', style='cfl')
params = {'head/' + k: v.astype(np.float64) for k, v in hp.items()}
for k, v in model.trunk.named().items(): params['conv/' + k] = v.astype(np.float64)
params['thr'] = np.float64(thr)
lcfg = O.LossCfg(reg_const=reg)
def oracle_loss(p, batch):
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    cp = {k.split('/', 1)[1].replace('/Conv/', '/').replace('biases', 'b'): v for k, v in tp.items() if k.startswith('conv/')}
    feats = [CO.convpcd_features(torch.clamp(torch.tensor(b, dtype=torch.float64), 0., 1.), shape, cp) for b in batch]
    head = {k.split('/', 1)[1]: v for k, v in tp.items() if k.startswith('head/')}
    total, _, _ = TO._torch_forward(cfg, lcfg, head, tp['thr'], tuple(feats))
    total = total + sum(0.5 * reg * (v * v).sum() for k, v in cp.items() if k.endswith('/V'))
    total.backward()
    return float(total.detach()), {k: (v.grad.numpy() if v.grad is not None else np.zeros_like(p[k])) for k, v in tp.items()}
batch = tuple(rng.rand(B, 784).astype(np.float32) * 1.2 - 0.1 for _ in range(4))
# oracle dF
tp = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
cp = {k.split('/', 1)[1].replace('/Conv/', '/').replace('biases', 'b'): v for k, v in tp.items() if k.startswith('conv/')}
feats = [CO.convpcd_features(torch.clamp(torch.tensor(b, dtype=torch.float64), 0., 1.), shape, cp) for b in batch]
for f in feats: f.retain_grad()
head = {k.split('/', 1)[1]: v for k, v in tp.items() if k.startswith('head/')}
total, _, _ = TO._torch_forward(cfg, lcfg, head, tp['thr'], tuple(feats))
total.backward()
# model pieces
eng = model.engine
x = torch.cat([model._pixels(batch[0]), model._pixels(batch[2]), model._pixels(batch[1]), model._pixels(batch[3])])
F = model.trunk.forward(x)
ref_F = torch.cat([feats[0], feats[2], feats[1], feats[3]]).detach()
print('F relerr', float((F.cpu().double() - ref_F).abs().max() / ref_F.abs().max()))
rows = (F[0:B], F[2 * B:3 * B], F[B:2 * B], F[3 * B:4 * B])
eng.fwd_bwd(rows)
dF = torch.empty_like(F)
ws = eng._workspace(B, 2)
H.pair_input_grad(eng.shape, eng.norm, B, eng.theta, ws, dF[0:2 * B], dF[2 * B:4 * B])
ref_dF = torch.cat([feats[0].grad, feats[2].grad, feats[1].grad, feats[3].grad])
err = (dF.cpu().double() - ref_dF).abs()
print('dF relerr', float(err.max() / ref_dF.abs().max()), 'argmax', np.unravel_index(int(err.argmax()), err.shape), 'max|ref|', float(ref_dF.abs().max()))
print('rows err', err.max(dim=1).values[:8], err.max(dim=1).values[12:16])
print('cols with err>1e-6:', int((err.max(dim=0).values > 1e-6 * float(ref_dF.abs().max())).sum()), 'of', err.shape[1])
model.trunk.backward(dF)
tg = model.trunk.named(model.trunk.grad)
ref_g = {k: v.grad.numpy() for k, v in tp.items() if k.startswith('conv/')}
# add reg term to oracle grads (total above excluded the conv reg)
for k, v in tg.items():
    r = ref_g['conv/' + k].copy()
    if k.endswith('/V'):
        r = r + reg * params['conv/' + k]
    e = np.abs(v - r)
    print('conv', k, 'max|ref|', np.abs(r).max(), 'relerr', e.max() / np.abs(r).max(), 'argmax', np.unravel_index(e.argmax(), e.shape), 'n>1e-5:', int((e > 1e-5 * np.abs(r).max()).sum()), 'of', e.size)
