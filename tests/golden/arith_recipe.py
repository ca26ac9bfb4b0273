"""Seeded recipe shared by the generator of the arithmetic goldens (make_arith_goldens.py, which runs
the REFERENCE's own model classes over tests/golden/tf_standin.py) and by the tests that replay the
same cases through oracle/ (CPU) and through the HIP path (GPU).

Only expected OUTPUTS are stored in arith_goldens.npz; every input (batches, initial variable values,
random draws z / eps / c) is regenerated from this recipe with legacy numpy RandomState streams, which
are stable across numpy versions.
"""
import zlib

import numpy as np

# keyword arguments of the reference's CFL(...) constructor (cfl/models/cfl.py:414-470) that a case may override
CFL_DEFAULTS = dict(
    is_double=False, disable_double=False, latent_shape=None, source_shape=None, ae_shape=None,
    data_norm=None, data_type='linear', num_components=2, pos_weight=None, latent_size=20, caffe_margin=None,
    gan=False, cgan=False, t_dim=None, dist_type='pcd', act_type=None, use_threshold=False, lr=1e-3, beta1=0.9,
    beta2=0.999, z_dim=20, z_stddev=1.0, g_dim=64, g_lr=2e-4, g_beta1=0.5, g_beta2=0.999, m_prj=None, m_enc=None,
    d_dim=64, d_lr=2e-4, d_beta1=0.5, d_beta2=0.999, lambda_gp=None, lambda_m=0.0, lambda_dra=0.5,
    directed=False, data_directed=False, model_type='linear', gan_type='conv', reg_const=0.0)

# normaliser flags of cfl.ops.dist_normalizer (cfl/ops.py:302-349) that a case may override
NORM_DEFAULTS = dict(data_scale=None, data_mean=None, latent_norm=None)

CASES = [
    # ---- `Dist` of cfl/models/dist.py (cfl.bin.train_dist) ----------------------------------------------
    dict(name='dist_k3', model='dist', input_shape=(24,), latent_size=5, num_components=3, batch_size=6,
         normalize_value=7.5, reg_const=0.0, steps=3, thr0=1e-6),
    dict(name='dist_k1_reg', model='dist', input_shape=(16,), latent_size=4, num_components=1, batch_size=5,
         normalize_value=2.0, reg_const=0.01, steps=3, thr0=0.4),
    # ---- `CFL`, linear encoders (cfl.bin.train --model-type linear) ----------------------------------------
    dict(name='cfl_pcd_pw', model='cfl', input_shape=(20,), latent_size=6, num_components=3, batch_size=8,
         dist_type='pcd', data_norm=[3.19], pos_weight=0.0625, use_threshold=True, steps=3, thr0=1e-6),
    dict(name='cfl_pcd_lm_tanh_reg', model='cfl', input_shape=(20,), latent_size=5, num_components=4, batch_size=6,
         dist_type='pcd', data_norm=[2.0], lambda_m=0.5, use_threshold=True, act_type='tanh', reg_const=5e-4,
         pos_weight=0.25, steps=3, thr0=0.7),
    dict(name='cfl_pcd_k1', model='cfl', input_shape=(12,), latent_size=7, num_components=1, batch_size=6,
         dist_type='pcd', use_threshold=True, steps=2, thr0=0.2),
    dict(name='cfl_monomer', model='cfl', input_shape=(18,), latent_size=5, num_components=3, batch_size=6,
         dist_type='monomer', use_threshold=True, pos_weight=0.25, act_type='sigmoid', reg_const=1e-3, steps=3,
         thr0=0.5),
    dict(name='cfl_siamese_caffe', model='cfl', input_shape=(16,), latent_size=9, num_components=1, batch_size=8,
         dist_type='siamese', caffe_margin=6.0, pos_weight=0.0625, use_threshold=False, data_norm=[1.7], steps=3,
         thr0=1e-6),
    dict(name='cfl_siamese_ut_relu', model='cfl', input_shape=(16,), latent_size=6, num_components=1, batch_size=6,
         dist_type='siamese', use_threshold=True, act_type='relu', steps=2, thr0=1.1),
    dict(name='cfl_pcd_directed', model='cfl', input_shape=(14,), latent_size=4, num_components=2, batch_size=6,
         dist_type='pcd', use_threshold=True, directed=True, data_directed=True, lambda_m=0.25, reg_const=1e-3,
         steps=3, thr0=0.3),
    dict(name='cfl_monomer_directed', model='cfl', input_shape=(14,), latent_size=4, num_components=2, batch_size=6,
         dist_type='monomer', use_threshold=True, directed=True, data_directed=True, steps=2, thr0=0.3),
    dict(name='cfl_pcd_tanh_data', model='cfl', input_shape=(10,), latent_size=4, num_components=2, batch_size=6,
         dist_type='pcd', use_threshold=True, data_type='tanh', data_mean=0.5, data_norm=[0.5], steps=2, thr0=0.3),
    # ---- `CFL`, conv encoder (cfl.bin.train --model-type conv; BASELINE config 0 family) ---------------------
    dict(name='cfl_conv_pcd', model='cfl', input_shape=(12, 12, 1), latent_size=6, num_components=2, batch_size=10,
         dist_type='pcd', use_threshold=True, data_type='sigmoid', model_type='conv', reg_const=5e-4, steps=2,
         thr0=0.3),
    # ---- MrCGAN post-epoch step (cfl.bin.train --gan) ------------------------------------------------------
    dict(name='gan_sr_double', model='cfl', gan_step=True, input_shape=(16, 16, 3), latent_shape=(12,),
         is_double=True, latent_size=6, num_components=2, batch_size=20, dist_type='pcd', use_threshold=True,
         data_type='tanh', data_mean=0.5, data_norm=[0.5], latent_norm=3.0, lambda_m=0.5, gan=True,
         gan_type='srgan', z_dim=4, lambda_gp=0.5, m_prj=0.2, m_enc=0.05, directed=False, data_directed=True,
         steps=2, thr0=0.3),
    dict(name='gan_conv_mnist', model='cfl', gan_step=True, input_shape=(16, 16, 1), latent_size=5,
         num_components=2, batch_size=20, dist_type='pcd', use_threshold=True, data_type='sigmoid',
         model_type='conv', lambda_m=0.5, gan=True, gan_type='conv', z_dim=3, lambda_gp=0.5, m_prj=0.5, m_enc=0.1,
         d_lr=1e-3, d_beta1=0.9, steps=2, thr0=0.3),
    dict(name='cgan_conv_t', model='cfl', gan_step=True, input_shape=(16, 16, 1), latent_size=5, num_components=2,
         batch_size=20, dist_type='pcd', use_threshold=True, data_type='sigmoid', model_type='conv', gan=True,
         cgan=True, t_dim=5, gan_type='conv', z_dim=3, lambda_gp=0.5, steps=2, thr0=0.3),
    dict(name='cgan_sr', model='cfl', gan_step=True, input_shape=(16, 16, 3), latent_shape=(12,), is_double=True,
         latent_size=6, num_components=2, batch_size=20, dist_type='pcd', use_threshold=True, data_type='tanh',
         data_mean=0.5, data_norm=[0.5], latent_norm=3.0, gan=True, cgan=True, gan_type='srgan', z_dim=4,
         lambda_gp=None, steps=1, thr0=0.3),
]


def case_by_name(name):
    for c in CASES:
        if c['name'] == name:
            return c
    raise KeyError(name)


def _seed(*parts):
    return zlib.crc32('|'.join(str(p) for p in parts).encode()) & 0x7fffffff


def prod(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def cfl_kwargs(case):
    """CFL(...) constructor arguments of a case (normalisers / transformers are added by the caller)."""
    kw = dict(CFL_DEFAULTS)
    for k in kw:
        if k in case:
            kw[k] = case[k]
    kw['input_shape'] = tuple(case['input_shape'])
    kw['batch_size'] = case['batch_size']
    return kw


def norm_kwargs(case):
    """arguments of cfl.ops.dist_normalizer"""
    kw = dict(NORM_DEFAULTS)
    for k in kw:
        if k in case:
            kw[k] = case[k]
    kw.update(input_shape=tuple(case['input_shape']), ae_shape=case.get('ae_shape'),
              data_norm=case.get('data_norm'), data_type=case.get('data_type', 'linear'))
    return kw


def _raw_data(rng, n, size, data_type, scale):
    """raw (un-normalised) rows whose normalised values straddle the clip bounds of `data_type`"""
    if data_type in ('sigmoid', 'tanh'):
        return rng.rand(n, size) * 1.2 - 0.1
    if data_type == 'relu':
        return rng.randn(n, size) * scale
    return rng.randn(n, size) * scale


def inputs(case, step):
    """Every input of one step: dict of float64 / int arrays.
    'batch' / 'val': pos_src, pos_dst, neg_src, neg_dst (cfl/input_data.py:585-589); for is_double each entry is an
    (image, latent) pair; 'unlabeled': (source, target) likewise; 'z', 'zs', 'eps', 'c' the random ops of the GAN."""
    rng = np.random.RandomState(_seed(case['name'], 'inputs', step))
    B = case['batch_size']
    size = prod(case['input_shape'])
    dt = case.get('data_type', 'linear')
    if case['model'] == 'dist':
        scale = case['normalize_value']
        mk = lambda: np.abs(rng.randn(B, size)) * scale / 2.0
        return dict(batch=[mk() for _ in range(4)], val=[mk() for _ in range(4)])
    scale = (case.get('data_norm') or [1.0])[0] if dt == 'linear' else 1.0
    double = case.get('is_double', False)
    lsize = prod(case['latent_shape']) if double else 0
    lscale = case.get('latent_norm') or 1.0

    def item():
        img = _raw_data(rng, B, size, dt, scale)
        if not double:
            return img
        return (img, rng.randn(B, lsize) * lscale)
    out = dict(batch=[item() for _ in range(4)], val=[item() for _ in range(4)])
    if case.get('gan'):
        # cfl/bin/train.py:30-40: source and target batches are drawn separately for directed data; otherwise ONE
        # unlabeled batch is fed to both
        if case.get('directed') or case.get('data_directed'):
            out['unlabeled'] = [item() for _ in range(2)]
        else:
            out['unlabeled'] = [item()] * 2
        K, zd = case['num_components'], case['z_dim']
        out['z'] = rng.randn(B, zd)
        out['zs'] = [rng.randn(B, zd) for _ in range(K)]
        out['eps'] = rng.rand(B, 1)
        out['c'] = rng.randint(0, K, size=(B,))
    return out


def init_value(case, name, shape):
    """Initial value of variable `name` (TF name without ':0').  Xavier-uniform weights as the reference
    initialises them; gains and biases are moved OFF their ones / zeros defaults so that every term of the
    weight-norm and bias arithmetic is exercised; the threshold starts at the case's `thr0` (1e-6 = the
    reference's init value, which sits exactly on the max(thr, 1e-6) tie)."""
    rng = np.random.RandomState(_seed(case['name'], 'init', name))
    leaf = name.rsplit('/', 1)[1]
    shape = tuple(int(s) for s in shape)
    if leaf == 'threshold':
        return np.float64(case.get('thr0', 1e-6))
    if leaf == 'g':
        return 1.0 + 0.2 * rng.randn(*shape)
    if leaf == 'biases':
        return 0.1 * rng.randn(*shape)
    if len(shape) == 2:
        fi, fo = shape
    else:
        rf = prod(shape[:-2])
        fi, fo = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=shape)


# ---- digests: small arrays are stored whole, large ones as a few seeded projections -----------------------
FULL_LIMIT = 4096


def digest(key, arr):
    """{suffix: array} to store / compare for `arr` under `key`"""
    a = np.asarray(arr, dtype=np.float64)
    if a.size <= FULL_LIMIT:
        return {key: a}
    flat = a.reshape(-1)
    rng = np.random.RandomState(_seed('probe', key))
    probes = rng.randn(4, flat.size)
    return {key + '#digest': np.concatenate([[flat.sum(), (flat * flat).sum()], probes @ flat, flat[:32]])}
