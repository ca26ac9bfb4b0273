"""The oracles against golden vectors captured from the REFERENCE'S OWN model classes.

tests/golden/make_arith_goldens.py imports cfl.models.dist.Dist / cfl.models.cfl.CFL (and through them
cfl.layers, cfl.ops, cfl.models.base, cfl.models.blocks) from the reference tree and runs their constructors
over an eager float64 stand-in for the TensorFlow primitives (tests/golden/tf_standin.py).  These tests
replay the same seeded cases through oracle/cfl_oracle.py, oracle/conv_oracle.py and oracle/gan_oracle.py
and require agreement to float64 round-off: every distance, score, loss part, accuracy, gradient (per
TensorFlow variable name) and every variable after 1-3 TF-Adam steps.

This pins what a restatement gets wrong -- which head feeds which side, reshape order k*L + l, softmax
axis, pos_weight / regulariser placement, which variables an optimiser owns, tie rules of maximum / relu --
to the reference's own lines.  It does not pin TensorFlow's kernels (the stand-in restates them from TF's
documentation); see DESIGN.md section 2.
"""
import numpy as np
import pytest

import tests.arith_goldens as G
from oracle import cfl_oracle as O

R = G.R

LINEAR = [c['name'] for c in R.CASES if c.get('model_type', 'linear') == 'linear' and not c.get('gan')]


def normalised(case, x):
    """the reference's data normaliser of the case (cfl/ops.py:198-202 for Dist, :66-124 for CFL)"""
    x = np.asarray(x, dtype=np.float64)
    if case['model'] == 'dist':
        return O.normalize(x, case['normalize_value'])
    lo, hi = {'sigmoid': (0., 1.), 'tanh': (-1., 1.), 'relu': (0., None), 'linear': (None, None)}[
        case.get('data_type', 'linear')]
    norm = (case.get('data_norm') or [None])[0]
    return O.normalize_v2(x, case.get('data_scale'), case.get('data_mean'), norm, lo, hi)


def oracle_cfgs(case):
    D = R.prod(case['input_shape'])
    style = 'dist' if case['model'] == 'dist' else 'cfl'
    cfg = O.EncoderCfg(D=D, L=case['latent_size'], K=case['num_components'],
                       dist_type=case.get('dist_type', 'pcd'), style=style, act_type=case.get('act_type'))
    if style == 'dist':
        lcfg = O.LossCfg(reg_const=case['reg_const'])
    else:
        lcfg = O.LossCfg(use_threshold=case.get('use_threshold', False), pos_weight=case.get('pos_weight'),
                         caffe_margin=case.get('caffe_margin'), lambda_m=case.get('lambda_m', 0.0),
                         reg_const=case.get('reg_const', 0.0))
    return cfg, lcfg


def tf_name_of(case, side, key):
    """inverse of G.split_encoder_name for this case"""
    for n, _ in G.meta(case['name'])['variables']:
        if G.split_encoder_name(n) == (side, key):
            return n
    raise KeyError((side, key))


@pytest.mark.parametrize('name', LINEAR)
def test_linear_models_match_reference_graph(name):
    case = R.case_by_name(name)
    cfg, lcfg = oracle_cfgs(case)
    src, dst, thr = G.encoder_params(G.initial_variables(case))
    tr = O.OracleTrainer(cfg, lcfg, lr=1e-3, dtype=np.float64, directed=dst is not None, params=src,
                         params_dst=dst)
    tr.raw_thr = np.float64(thr)
    m = G.meta(name)
    owner = {v: t for t, vs in m['optimizers'].items() for v in vs}
    nograd = {v for vs in m.get('no_gradient', {}).values() for v in vs}
    thr_name = [n for n, _ in m['variables'] if n.endswith('threshold/threshold')][0]
    for step in range(case['steps']):
        inp = R.inputs(case, step)
        batch = tuple(normalised(case, x) for x in inp['batch'])
        vbatch = tuple(normalised(case, x) for x in inp['val'])
        sc, g, gd, dthr, dthr_aux = O.train_step_loss_and_grads(cfg, lcfg, tr.params, tr.raw_thr, batch,
                                                               tr.params_dst)
        chk = lambda k, a: G.check(case, step, k, a)
        chk('s_pos_dists', sc['d_pos'][:, None])
        chk('s_neg_dists', sc['d_neg'][:, None])
        chk('s_total_loss', sc['total'])
        chk('s_loss_reg', sc['reg'])
        chk('thres_loss' if case['model'] == 'dist' else 's_thres_loss', sc['thres'])
        chk('s_p_loss_pos', sc['loss_pos'])
        chk('s_p_loss_neg', sc['loss_neg'])
        chk('s_accuracy', sc['accuracy'])
        chk('s_margins', sc['margins'])
        if G.has(case, step, 's_cd_loss'):
            chk('s_cd_loss', sc['cd'])
        chk('threshold', max(tr.raw_thr, 1e-6))
        chk('val_pos_scores', tr.scores(vbatch[0], vbatch[1])[:, None])
        chk('val_neg_scores', tr.scores(vbatch[2], vbatch[3])[:, None])
        vacc = 0.5 * ((tr.scores(vbatch[0], vbatch[1]) > 0).mean() + (tr.scores(vbatch[2], vbatch[3]) <= 0).mean())
        chk('val_s_accuracy', vacc)
        # gradients, keyed by the reference's variable names and by the optimiser that owns the variable
        for side, grads in (('src', g), ('dst', gd)):
            if grads is None:
                continue
            for k, a in grads.items():
                n = tf_name_of(case, side, k)
                if n in nograd:
                    assert not np.any(a), n          # TF: gradient None -> the variable is left alone
                    continue
                chk('grad/%s/%s' % (owner[n], n), a)
        chk('grad/%s/%s' % (owner[thr_name], thr_name), dthr if lcfg.use_threshold else dthr_aux)
        tr.step(batch)
        for side, p in (('src', tr.params), ('dst', tr.params_dst)):
            if p is None:
                continue
            for k, a in p.items():
                chk('after/' + tf_name_of(case, side, k), a)
        chk('after/' + thr_name, tr.raw_thr)


def test_variable_names_and_optimizer_ownership():
    """SURVEY App. D was derived by reading scope code; here the names come out of the reference's own
    variable_scope / get_variable calls."""
    m = G.meta('dist_k3')
    assert [n for n, _ in m['variables']] == [
        'Dist/Encoder/latent_outputs/fully_connected/weights', 'Dist/Encoder/latent_outputs/fully_connected/biases',
        'Dist/Encoder/pcd_outputs/fully_connected/weights', 'Dist/Encoder/pcd_outputs/fully_connected/biases',
        'Dist/Thresholder/threshold/threshold']
    assert m['model_name'] == 'linear_dist_ls_5_nc_3_reg_0.0_norm_7.5'
    m = G.meta('cfl_siamese_caffe')
    # siamese: one head, no bias (cfl/models/base.py:45-46); threshold trained by its own Adam when --use-threshold is off
    assert m['optimizers'] == {
        's_optim': ['CFL/DistEncoder/outputs/fully_connected/g', 'CFL/DistEncoder/outputs/fully_connected/V'],
        'th_optim': ['CFL/Thresholder/threshold/threshold']}
    m = G.meta('cfl_monomer')
    names = [n for n, _ in m['variables']]
    assert 'CFL/DistEncoder/monomer_outputs/fully_connected/V' in names
    assert not any(n.endswith('biases') for n in names)        # monomer heads carry no biases
    m = G.meta('gan_sr_double')
    names = [n for n, _ in m['variables']]
    assert 'CFL/Discriminator/conv2/Conv_4/V' in names and 'CFL/Generator/subpixel_block1/Conv/V' in names
    assert all(n.startswith('CFL/Discriminator/') for n in m['optimizers']['post_d_optim'])
    assert all(n.startswith('CFL/Generator/') for n in m['optimizers']['post_g_optim'])


# ---------------------------------------------------------------------------------------------------------
# conv encoder and MrCGAN cases: torch float64 oracles (oracle/conv_oracle.py, oracle/gan_oracle.py)
# ---------------------------------------------------------------------------------------------------------
def _t(a):
    import torch
    return torch.tensor(np.asarray(a, dtype=np.float64))


def conv_trunk_params(values, scope):
    """{'conv1/V': .., 'conv1/g': .., 'conv1/b': ..} of encoder `scope` ('CFL/DistEncoder') in conv_oracle naming"""
    out = {}
    for n, v in values.items():
        parts = n.split('/')
        if n.startswith(scope + '/conv') and parts[3] == 'Conv':
            out[parts[2] + '/' + {'V': 'V', 'g': 'g', 'biases': 'b'}[parts[4]]] = v
    return out


def encoder_acts(case, values, x_norm, side='src'):
    """(activations [B,L], prototype activations [B,K,L] | None) of one encoder on normalised rows, float64 numpy:
    heads of cfl/models/base.py:43-92 on the trunk features (identity for --model-type linear)."""
    import torch
    from oracle import conv_oracle as CO
    src, dst, _ = G.encoder_params(values)
    p = dst if (side == 'dst' and dst is not None) else src
    x = np.asarray(x_norm, dtype=np.float64)
    if case.get('model_type') == 'conv':
        cp = {k: _t(v) for k, v in conv_trunk_params(values, 'CFL/DistEncoder').items()}
        x = CO.convpcd_features(_t(x), tuple(case['input_shape']), cp).numpy()
    ya, _ = O.fc_head(x, p, 'outputs', True)
    acts = O._act(ya, case.get('act_type'))
    protos = None
    if 'proto/W' in p:
        yp, _ = O.fc_head(x, p, 'proto', True)
        protos = O._act(yp, case.get('act_type')).reshape(x.shape[0], case['num_components'], case['latent_size'])
    return acts, protos


def ae_normalised(case, x):
    """ae_normalizer == data_normalizer when ae_shape is the input shape (cfl/ops.py:335-347)"""
    return normalised(case, x)


def latent_normalised(case, x):
    ln = case.get('latent_norm')
    return np.asarray(x, dtype=np.float64) / ln if ln else np.asarray(x, dtype=np.float64)


def test_conv_encoder_matches_reference_graph():
    import torch
    import tests.test_oracle as TO
    from oracle import conv_oracle as CO
    case = R.case_by_name('cfl_conv_pcd')
    cfg, lcfg = oracle_cfgs(case)
    shape = tuple(case['input_shape'])
    values = G.initial_variables(case)
    cfg.D = CO.convpcd_layers(shape)[1]
    adam = O.AdamState(1e-3)
    reg = case['reg_const']
    thr_name = 'CFL/Thresholder/threshold/threshold'
    for step in range(case['steps']):
        inp = R.inputs(case, step)
        tv = {n: torch.tensor(v, requires_grad=True) for n, v in values.items()}
        cp = conv_trunk_params(tv, 'CFL/DistEncoder')
        feats = tuple(CO.convpcd_features(_t(normalised(case, x)), shape, cp) for x in inp['batch'])
        head = {G.split_encoder_name(n)[1]: v for n, v in tv.items() if G.split_encoder_name(n)}
        total, lp, ln = TO._torch_forward(cfg, lcfg, head, tv[thr_name], feats)
        total = total + sum(0.5 * reg * (v * v).sum() for k, v in cp.items() if k.endswith('/V'))
        G.check(case, step, 's_total_loss', total.item())
        G.check(case, step, 's_p_loss_pos', lp.item())
        G.check(case, step, 's_p_loss_neg', ln.item())
        total.backward()
        grads = {}
        for n, v in tv.items():
            G.check(case, step, 'grad/s_optim/' + n, v.grad.numpy())
            grads[n] = v.grad.numpy()
        adam.apply(values, grads)
        for n, v in values.items():
            G.check(case, step, 'after/' + n, v)


GAN_CASES = [c['name'] for c in R.CASES if c.get('gan')]


def _gan_oracle(case, values):
    import torch
    from oracle import gan_oracle as GO
    ae_shape = tuple(case['input_shape'])
    kw = dict(d_lr=case.get('d_lr', 2e-4), d_beta1=case.get('d_beta1', 0.5), g_lr=case.get('g_lr', 2e-4),
              g_beta1=case.get('g_beta1', 0.5), lambda_gp=case.get('lambda_gp'), lambda_dra=0.5)
    if case.get('cgan'):
        c_dim = (R.prod(case['latent_shape']) if case.get('is_double') else R.prod(case['input_shape'])) \
            if case.get('t_dim') else case['latent_size']
        go = GO.GanOracle(case['gan_type'], ae_shape, case['data_type'], case['z_dim'], case['latent_size'],
                          cgan=True, c_dim=c_dim, t_dim=case.get('t_dim'), **kw)
    else:
        go = GO.GanOracle(case['gan_type'], ae_shape, case['data_type'], case['z_dim'], case['latent_size'],
                          m_enc=case.get('m_enc'), m_prj=case.get('m_prj'), **kw)
    gp = {n[len('CFL/Generator/'):]: torch.tensor(v) for n, v in values.items() if n.startswith('CFL/Generator/')}
    dp = {n[len('CFL/Discriminator/'):]: torch.tensor(v) for n, v in values.items()
          if n.startswith('CFL/Discriminator/')}
    assert set(gp) == set(go.gp) and set(dp) == set(go.dp), (set(gp) ^ set(go.gp), set(dp) ^ set(go.dp))
    go.gp, go.dp = gp, dp
    go.g_adam = GO.AdamTF(gp, kw['g_lr'], kw['g_beta1'])
    go.d_adam = GO.AdamTF(dp, kw['d_lr'], kw['d_beta1'])
    return go


def gan_batch(case, values, inp):
    """the oracle's inputs of one post-epoch step, derived from the recipe inputs the way CFL._build_model wires
    them (cfl/models/cfl.py:706-806)"""
    double = case.get('is_double', False)
    img = lambda item: item[0] if double else item
    enc_in = lambda item: latent_normalised(case, item[1]) if double else normalised(case, item)
    z, eps, c = inp['z'], inp['eps'], inp['c']
    rows = np.arange(case['batch_size'])
    if case.get('cgan'):
        ps, pt, ns, nt = inp['batch']
        if case.get('t_dim'):
            pos_c, neg_c = enc_in(ps), enc_in(ns)
        else:
            pos_c = encoder_acts(case, values, enc_in(ps))[0]
            neg_c = encoder_acts(case, values, enc_in(ns))[0]
        return (ae_normalised(case, img(pt)), ae_normalised(case, img(nt)), pos_c, neg_c, z, eps)
    us, ut = inp['unlabeled']
    enc_act = encoder_acts(case, values, enc_in(ut), 'dst')[0]
    prj_c = encoder_acts(case, values, enc_in(us), 'src')[1][rows, c]
    neg_c = encoder_acts(case, values, enc_in(inp['batch'][2]), 'src')[1][rows, c]
    neg_tgt = encoder_acts(case, values, enc_in(inp['batch'][3]), 'dst')[0]
    return (ae_normalised(case, img(ut)), enc_act, prj_c, neg_c, neg_tgt, z, eps)


@pytest.mark.parametrize('name', GAN_CASES)
def test_gan_step_matches_reference_graph(name):
    case = R.case_by_name(name)
    values = G.initial_variables(case)
    go = _gan_oracle(case, values)
    for step in range(case['steps']):
        inp = R.inputs(case, step)
        batch = gan_batch(case, values, inp)
        chk = lambda k, a: G.check(case, step, k, a, rtol=1e-8, atol=1e-10)
        if not case.get('cgan'):
            for k, a in zip(('enc_act', 'prj_c', 'neg_c', 'neg_tgt_act'), batch[1:5]):
                chk(k, a)                 # the encoder side of the GAN wiring
        else:
            chk('pos_c', batch[2])
            chk('neg_c', batch[3])
        tb = tuple(_t(b) for b in batch)
        d_total, g_total, parts, d_grads, g_grads = go.losses_and_grads(*tb)
        chk('d_total_loss', d_total.item())
        chk('g_total_loss', g_total.item())
        for k, v in parts.items():
            if G.has(case, step, k):
                chk(k, v.numpy() if hasattr(v, 'numpy') else v)
        if 'g_activations' in parts:
            chk('g_activations', parts['g_activations'].numpy())
        nograd = {v for vs in G.meta(name).get('no_gradient', {}).values() for v in vs}
        for k, g in d_grads.items():
            n = 'CFL/Discriminator/' + k
            if n in nograd:
                assert not bool(g.abs().max() > 0), n
            else:
                chk('grad/post_d_optim/' + n, g.numpy())
        for k, g in g_grads.items():
            chk('grad/post_g_optim/CFL/Generator/' + k, g.numpy())
        go.step(*tb)
        for k, v in go.dp.items():
            chk('after/CFL/Discriminator/' + k, v.numpy())
        for k, v in go.gp.items():
            chk('after/CFL/Generator/' + k, v.numpy())


# ---------------------------------------------------------------------------------------------------------
# the HIP path against the same goldens (-m gpu): the product's host classes (cfl.models.dist.Dist,
# cfl.models.cfl.CFL), loaded with the recipe's initial variables through their TensorFlow-named checkpoint
# state, fed the recipe's batches; every call goes through the C ABI of libcfl_hip.so.
# ---------------------------------------------------------------------------------------------------------
ALL_CASES = [c['name'] for c in R.CASES]
F32_TOL = 2e-5          # fp32 kernels against the float64 goldens, relative to max(1, |golden|)


def build_product_model(case):
    from cfl import ops
    if case['model'] == 'dist':
        from cfl.models.dist import construct_model
        nv = case['normalize_value']
        model, _ = construct_model(tuple(case['input_shape']), case['latent_size'], case['num_components'], 1e-3, 0.9,
                                   0.999, case['batch_size'], nv, reg_const=case['reg_const'],
                                   data_normalizer=ops.normalizer(nv, 0.), data_unnormalizer=ops.unnormalizer(nv, 0.))
        return model
    from cfl.models.cfl import construct_model
    kw = R.cfl_kwargs(case)
    tr, vtr = ops.dist_transformer(source_shape=None, input_shape=kw['input_shape'], data_random_crop=False,
                                   data_mirror=False)
    dn, dun, aen, aeun, ln = ops.dist_normalizer(**R.norm_kwargs(case))
    model, _ = construct_model(train_data_transformer=tr, val_data_transformer=vtr,
                               ae_transformer=ops.dist_ae_transformer(kw['input_shape'], kw['ae_shape']),
                               data_normalizer=dn, data_unnormalizer=dun, ae_normalizer=aen, ae_unnormalizer=aeun,
                               latent_normalizer=ln, seed=0, **kw)
    return model


def load_initial(model, case):
    st = model.checkpoint_state()
    init = G.initial_variables(case)
    assert set(st['variables']) == set(init), sorted(set(st['variables']) ^ set(init))
    for n, v in init.items():
        assert tuple(np.shape(st['variables'][n])) == tuple(np.shape(v)), n
        st['variables'][n] = np.asarray(v, np.float32)
        st['adam_m'][n] = np.zeros_like(st['variables'][n])
        st['adam_v'][n] = np.zeros_like(st['variables'][n])
    model.load_checkpoint_state(st)
    assert model.get_name() == G.meta(case['name'])['model_name']


def labeled(case, items):
    """a labeled batch in cfl/input_data.py:581-589 order (image and latent interleaved for double data)"""
    if case.get('is_double'):
        out = []
        for img, lat in items:
            out += [img.astype(np.float32), lat.astype(np.float32)]
        return out
    return [x.astype(np.float32) for x in items]


def unl(case, item):
    return [a.astype(np.float32) for a in item] if case.get('is_double') else [item.astype(np.float32)]


class _Diffs(object):
    def __init__(self, case):
        self.case, self.bad, self.worst = case, [], {}

    def scalar(self, step, key, got, tol=F32_TOL):
        if not G.has(self.case, step, key):
            return
        ref = float(G.expected(self.case, step, key))
        err = abs(float(got) - ref) / max(1.0, abs(ref))
        self.worst[key] = max(self.worst.get(key, 0.0), err)
        if not err <= tol:
            self.bad.append('step %d %s: got %.8g want %.8g (rel %.2e > %.1e)' % (step, key, got, ref, err, tol))

    def array(self, step, key, got, tol=F32_TOL, frac=1.0, digest_tol=3e-4):
        """small tensors element-wise (a fraction `frac` of the entries within tol of the tensor's scale), large
        ones through their seeded projections"""
        full = '%s/step%d/%s' % (self.case['name'], step, key)
        got = np.asarray(got, dtype=np.float64)
        if full in G.npz().files:
            ref = G.npz()[full]
            if got.shape != ref.shape:
                self.bad.append('%s: shape %r vs %r' % (full, got.shape, ref.shape))
                return
            scale = max(float(np.abs(ref).max()), 1e-3)
            d = np.abs(got - ref) / scale
            ok = float((d <= tol).mean()) if d.size else 1.0
            self.worst[key] = max(self.worst.get(key, 0.0), float(d.max()) if d.size else 0.0)
            if ok < frac:
                self.bad.append('%s: %.4f of entries within %.1e (max %.2e)' % (full, ok, tol, d.max()))
            return
        for k, mine in R.digest(full, got).items():
            ref = G.npz()[k]
            # [sum, sumsq, 4 probes, 32 leading entries]: relative to the size of the probe projections
            scale = max(float(np.abs(ref[2:6]).max()), 1e-6)
            err = float(np.abs(mine[2:] - ref[2:]).max()) / scale
            self.worst[key] = max(self.worst.get(key, 0.0), err)
            if not err <= digest_tol:
                self.bad.append('%s: digest differs by %.2e of its scale' % (k, err))

    def finish(self):
        assert not self.bad, '\n'.join(self.bad[:40]) + '\nworst: %r' % (sorted(self.worst.items(), key=lambda kv: -kv[1])[:8],)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ALL_CASES)
def test_hip_matches_reference_graph(name):
    """HIP (fp32) against the float64 goldens of the reference's own graph code: step 0 is a pure function of the
    inputs (held to 2e-5); later steps follow the fp32 trajectory (losses 1e-5 for the pair step; conv / GAN stacks
    are held to the tolerance their kinked activations allow, see tests/test_activation_masks_gpu.py)."""
    import torch
    from cfl import hipabi as H
    case = R.case_by_name(name)
    model = build_product_model(case)
    load_initial(model, case)
    m = G.meta(name)
    df = _Diffs(case)
    kinked = case.get('model_type') == 'conv' or case.get('gan')
    for step in range(case['steps']):
        inp = R.inputs(case, step)
        later = step > 0
        if not case.get('gan_step'):
            val = labeled(case, inp['val'])
            vsel = model.select_batch(val) if hasattr(model, 'select_batch') else val
            df.array(step, 'val_pos_scores', model.predict(vsel[0], vsel[1]), tol=5e-5 if kinked and later else F32_TOL)
            df.array(step, 'val_neg_scores', model.predict(vsel[2], vsel[3]), tol=5e-5 if kinked and later else F32_TOL)
            model.train_step(labeled(case, inp['batch']))
            s = model.scalars()
            tol = (5e-4 if kinked else 1e-5) if later else F32_TOL
            df.scalar(step, 's_total_loss', s['total'], tol)
            df.scalar(step, 's_loss_reg', s['reg'], tol)
            df.scalar(step, 'thres_loss' if case['model'] == 'dist' else 's_thres_loss', s['thres'], tol)
            df.scalar(step, 's_p_loss_pos', s['loss_pos'], tol)
            df.scalar(step, 's_p_loss_neg', s['loss_neg'], tol)
            df.scalar(step, 's_cd_loss', s['cd'], tol)
            df.scalar(step, 's_accuracy', s['accuracy'], 1e-6 if not later else 0.5 / case['batch_size'] + 1e-6)
            df.scalar(step, 's_margins', s['mean_d_pos'] - s['mean_d_neg'], tol)
            df.scalar(step, 'threshold', s['threshold'], tol)
            df.scalar(step, 's_pos_dists_adapt', s['dist_adapt_pos'], tol)
            df.scalar(step, 's_neg_dists_adapt', s['dist_adapt_neg'], tol)
            # gradients of the fused pair step by TensorFlow variable name
            if step == 0:
                owner = {v: t for t, vs in m['optimizers'].items() for v in vs}
                nograd = {v for vs in m.get('no_gradient', {}).values() for v in vs}
                named_grad = model._named(model.engine.grad)
                for n, g in named_grad.items():
                    if n in nograd or n not in owner:
                        continue
                    df.array(step, 'grad/%s/%s' % (owner[n], n), g, tol=3e-5)
                if getattr(model, 'trunk', None) is not None:
                    for n, g in model.trunk.named(model.trunk.grad).items():
                        df.array(step, 'grad/s_optim/CFL/DistEncoder/' + n, g, tol=1e-4, frac=0.999, digest_tol=1e-3)
        else:
            lab = labeled(case, inp['batch'])
            draws = (inp['z'], inp['eps'], inp['c'])
            if case.get('cgan'):
                model.post_step(lab, draws=draws)
            else:
                model.post_step(lab, unl(case, inp['unlabeled'][0]), unl(case, inp['unlabeled'][1]), draws=draws)
            s = model.gan_phase.read_scalars()
            tol = 2e-3 if later else 5e-5
            for k in ('d_total_loss', 'g_total_loss', 'd_loss_real', 'd_loss_fake', 'd_loss_neg', 'd_grad_loss',
                      'd_loss_d', 'g_loss', 'g_loss_d', 'g_loss_d_neg', 'g_loss_int'):
                if k in s:
                    df.scalar(step, k, s[k], tol)
            if step == 0:
                for k in ('d_real_accuracy', 'd_fake_accuracy', 'g_accuracy'):
                    df.scalar(step, k, s[k], 1e-6)
                ph = model.gan_phase
                nograd = {v for vs in m.get('no_gradient', {}).values() for v in vs}
                for tag, net in (('post_d_optim', ph.disc), ('post_g_optim', ph.gen)):
                    for n, g in net.pool.named(net.pool.grad).items():
                        if 'CFL/' + n not in nograd:
                            df.array(step, 'grad/%s/CFL/%s' % (tag, n), g, tol=3e-4, frac=0.995, digest_tol=1e-3)
        # variables after the step (Adam's first steps move every entry by about lr whatever the size of its
        # gradient, so entries whose gradient is at the fp32 noise level may go the other way: a fraction, not all)
        vars_now = model.checkpoint_state()['variables']
        lr = max(case.get('d_lr', 2e-4), case.get('g_lr', 2e-4)) if case.get('gan_step') else 1e-3
        for n, v in vars_now.items():
            key = 'after/' + n
            full = '%s/step%d/%s' % (name, step, key)
            if full in G.npz().files:
                ref = G.npz()[full]
                d = np.abs(np.asarray(v, np.float64) - ref)
                if d.size and not (float((d <= 1e-5 + 0.05 * lr).mean()) >= 0.97 and d.max() <= 2.5 * (step + 1) * lr):
                    df.bad.append('%s: %.4f within tol, max %.2e' % (full, float((d <= 1e-5 + 0.05 * lr).mean()), d.max()))
            else:
                df.array(step, key, v, digest_tol=2e-3)
    df.finish()


@pytest.mark.gpu
def test_padded_width_with_shift_normaliser_roundtrip():
    """A feature width that is not a multiple of 64 together with a normaliser that maps 0 to a non-zero value
    (--data-type tanh --data-mean .5 --data-norm .5: add = -1): the zero padding must stay out of the model.  The
    padded weight rows never train, so a checkpoint (which stores the true [D, N] variables) restores the very
    model that was trained.  (The multi-step parity of this configuration is the golden case cfl_pcd_tanh_data.)"""
    import torch
    from cfl import hipabi as H
    case = dict(R.case_by_name('cfl_pcd_tanh_data'), input_shape=(200,), batch_size=32, latent_size=8)
    model = build_product_model(case)
    rng = np.random.RandomState(3)
    mk = lambda: [rng.rand(32, 200).astype(np.float32) * 1.2 - 0.1 for _ in range(4)]
    for _ in range(4):
        model.train_step(mk())
    p, _, _ = H.unpack_theta(model.engine.shape, model.engine.theta)
    for k in ('outputs/W', 'proto/W'):
        assert not p[k][200:].any(), k              # rows under the zero padding received no update
        assert np.abs(p[k][:200]).max() > 0
    twin = build_product_model(case)
    twin.load_checkpoint_state(model.checkpoint_state())
    assert torch.equal(twin.engine.theta, model.engine.theta)
    probe = mk()
    assert np.array_equal(twin.predict(probe[0], probe[1]), model.predict(probe[0], probe[1]))
    nxt = mk()
    model.train_step(nxt)
    twin.train_step(nxt)
    assert torch.equal(twin.engine.theta, model.engine.theta)
