"""Generate the ARITHMETIC golden vectors by running the reference's own model classes.

Run in the build container only (the reference tree never travels):

    python tests/golden/make_arith_goldens.py

What runs: `/root/reference/cfl/{ops,layers}.py` and `/root/reference/cfl/models/{base,blocks,dist,cfl}.py`,
imported unmodified, with `tests/golden/tf_standin.py` registered as `tensorflow`.  The reference's
`Dist(...)` / `CFL(...)` constructors build their whole graph (encoders, distances, thresholder, losses,
optimisers, GAN stacks, gradient penalty, summaries) from their own lines; the stand-in evaluates every
TF primitive eagerly in float64 (torch autograd behind `tf.gradients` / `minimize`).  One training step
== `sess.run([s_optim, update_stats(, th_optim)])` (cfl/models/cfl.py:1399-1414, cfl/bin/train_dist.py:81-82)
or `sess.run([post_d_optim, post_g_optim])` (cfl/models/cfl.py:1487-1497).

PINNED by this: the composition of the graph (see tf_standin.py).  NOT pinned: TensorFlow's kernels --
the primitives are restated from TF's documentation, so parity with a real TF run remains unverified.

Stored: tests/golden/arith_goldens.npz (expected outputs only; large tensors as seeded projections) and
tests/golden/arith_goldens_meta.json (per case: variable names / shapes in creation order, optimiser
ownership, model name).  Inputs are regenerated from tests/golden/arith_recipe.py.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.path.insert(0, HERE)

import arith_recipe as R  # noqa: E402
import tf_standin as S  # noqa: E402


def data(a):
    """an input batch: requires_grad so that tf.gradients(D(X_hat), X_hat) has a path (cfl/models/cfl.py:987)"""
    if a is None:
        return None
    return S.Tensor(torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True))


def flat_batches(case, items):
    """order of the `batches` list CFL._build_train_data / _build_val_data index (cfl/models/cfl.py:96-134)"""
    if case.get('is_double'):
        out = []
        for img, lat in items:
            out += [data(img), data(lat)]
        return out
    return [data(x) for x in items]


def unlabeled_batches(case, items):
    """cfl/models/cfl.py:197-224: non-double uses entries 0 and 2, double 0, 1 and 3, 4 (labels in between)"""
    if case.get('is_double'):
        (si, sl), (ti, tl) = items
        return [data(si), data(sl), None, data(ti), data(tl), None]
    s, t = items
    return [data(s), None, data(t), None]


def build(case, step, ref):
    """One construction of the reference model on the inputs of `step`."""
    S.reset_graph()
    inp = R.inputs(case, step)
    st = S.state()
    st.feeds = {}
    if case['model'] == 'dist':
        nv = case['normalize_value']
        model = ref['dist'].Dist(
            input_shape=tuple(case['input_shape']), latent_size=case['latent_size'],
            num_components=case['num_components'], batch_size=case['batch_size'], lr=1e-3, beta1=0.9,
            beta2=0.999, batches=[data(x) for x in inp['batch']], val_batches=[data(x) for x in inp['val']],
            normalize_value=nv, data_normalizer=ref['ops'].normalizer(nv, 0.),
            data_unnormalizer=ref['ops'].unnormalizer(nv, 0.), reg_const=case['reg_const'])
        return model, inp
    kw = R.cfl_kwargs(case)
    nk = R.norm_kwargs(case)
    ops = ref['ops']
    tr, vtr = ops.dist_transformer(source_shape=None, input_shape=kw['input_shape'], data_random_crop=False,
                                   data_mirror=False)
    ae_tr = ops.dist_ae_transformer(input_shape=kw['input_shape'], ae_shape=kw['ae_shape'])
    dn, dun, aen, aeun, ln = ops.dist_normalizer(**nk)
    if case.get('gan'):
        st.feeds.update(z=inp['z'], eps=inp['eps'], c=inp['c'])
        for i, z in enumerate(inp['zs']):
            st.feeds['z%d' % i] = z
    model = ref['cfl'].CFL(
        batches=flat_batches(case, inp['batch']), val_batches=flat_batches(case, inp['val']),
        unlabeled_batches=unlabeled_batches(case, inp['unlabeled']) if case.get('gan') else
        unlabeled_batches(case, [inp['batch'][0], inp['batch'][1]]),
        train_data_transformer=tr, val_data_transformer=vtr, ae_transformer=ae_tr, data_normalizer=dn,
        data_unnormalizer=dun, ae_normalizer=aen, ae_unnormalizer=aeun, latent_normalizer=ln, **kw)
    return model, inp


def val(t):
    return np.asarray(t.numpy() if isinstance(t, S.Tensor) else t, dtype=np.float64)


DIST_SCALARS = ('s_total_loss', 's_loss_reg', 's_thres_loss', 'thres_loss', 's_p_loss_pos', 's_p_loss_neg',
                's_cd_loss', 's_cd_loss_pos', 's_cd_loss_neg', 's_accuracy', 'val_s_accuracy', 's_margins',
                's_pos_dists_adapt', 's_neg_dists_adapt', 's_margin_adapt')
GAN_SCALARS = ('d_total_loss', 'g_total_loss', 'd_loss', 'd_loss_real', 'd_loss_fake', 'd_loss_enc', 'd_loss_prj',
               'd_loss_neg', 'd_grad_loss', 'd_loss_d', 'g_loss', 'g_loss_enc', 'g_loss_prj', 'g_loss_d',
               'g_loss_d_neg', 'g_loss_int', 'd_real_accuracy', 'd_fake_accuracy', 'g_accuracy')


def record_step(case, model, out, pre):
    put = lambda k, a: out.update(R.digest(pre + k, val(a)))
    for name in DIST_SCALARS:
        if hasattr(model, name):
            put(name, getattr(model, name))
    put('s_pos_dists', model.s_pos_dists)
    put('s_neg_dists', model.s_neg_dists)
    put('s_pos_scores', model.s_pos_predicts.outputs)
    put('s_neg_scores', model.s_neg_predicts.outputs)
    put('val_pos_scores', model.val_s_pos_predicts.outputs)
    put('val_neg_scores', model.val_s_neg_predicts.outputs)
    put('threshold', model.s_pos_predicts.threshold)
    if case.get('gan'):
        for name in GAN_SCALARS:
            if hasattr(model, name):
                put(name, getattr(model, name))
        put('g_activations', model.g.activations)
        put('d_real_logits', model.d_real.disc_outputs)
        put('d_fake_logits', model.d_fake.disc_outputs)
        if hasattr(model.d_real, 'latent_activations'):
            put('d_real_latent', model.d_real.latent_activations)
        if model.lambda_gp:
            put('X_hat', model.X_hat)
        if not case.get('cgan'):
            # what the encoders hand to the GAN (inputs of oracle/gan_oracle.py)
            put('enc_act', model.s_encoder.activations)
            put('prj_c', model.s_g.one_prototype_activations)
            put('neg_c', model.s_neg_src.one_prototype_activations)
            put('neg_tgt_act', model.s_neg_target.activations)
        else:
            put('pos_c', model.data_pos_source_latent if model.t_dim else model.s_pos_src.activations)
            put('neg_c', model.data_neg_source_latent if model.t_dim else model.s_neg_src.activations)
            put('g_int_activations', model.g_int.activations)


def run_case(case, ref):
    S.reset_all(0)
    out, meta = {}, dict(name=case['name'])
    # dry construction: learn the variable names / shapes the reference creates, then set the recipe's values
    model, _ = build(case, 0, ref)
    st = S.state()
    names = list(st.created)
    shapes = {n: list(st.values[n].shape) for n in names}
    for n in names:
        st.values[n] = np.asarray(R.init_value(case, n, shapes[n]), dtype=np.float64).reshape(shapes[n])
    meta['variables'] = [[n, shapes[n]] for n in names]
    meta['model_name'] = model.get_name()
    gan_step = bool(case.get('gan_step'))
    for step in range(case['steps']):
        model, _ = build(case, step, ref)
        pre = '%s/step%d/' % (case['name'], step)
        record_step(case, model, out, pre)
        if case['model'] == 'dist':
            ops, tags = [model.s_optim], ['s_optim']
            ema = []
        elif gan_step:
            ops, tags = [model.post_d_optim, model.post_g_optim], ['post_d_optim', 'post_g_optim']
            ema = []
        else:
            ops, tags = [model.s_optim], ['s_optim']
            if not model.use_threshold:
                ops.append(model.th_optim)
                tags.append('th_optim')
            ema = model.update_stats
        if step == 0:
            meta['optimizers'] = {t: [v.var_name for v in op.var_list] for t, op in zip(tags, ops)}
        grads = S.run_train_ops(ops)
        S.run_ema_ops(ema)
        for t, g in zip(tags, grads):
            for n, a in g.items():
                if a is None:
                    meta.setdefault('no_gradient', {}).setdefault(t, [])
                    if n not in meta['no_gradient'][t]:
                        meta['no_gradient'][t].append(n)
                    continue
                out.update(R.digest(pre + 'grad/' + t + '/' + n, a))
        for n in names:
            out.update(R.digest(pre + 'after/' + n, st.values[n]))
        if ema:
            # the averages the progress bar shows (cfl/models/cfl.py:1405-1414): read after the update
            model2_keys = sorted(st.ema)
            out[pre + 'ema'] = np.array([float(st.ema[k]) for k in model2_keys])
    return out, meta


def main():
    S.install()
    sys.path.insert(0, REF)
    import cfl.ops as rops
    import cfl.models.dist as rdist
    import cfl.models.cfl as rcfl
    ref = dict(ops=rops, dist=rdist, cfl=rcfl)
    only = set(sys.argv[1:])
    allout, metas = {}, []
    for case in R.CASES:
        if only and case['name'] not in only:
            continue
        out, meta = run_case(case, ref)
        allout.update(out)
        metas.append(meta)
        print('%-24s %-70s %4d arrays' % (case['name'], meta['model_name'], len(out)))
    if only:
        return
    np.savez_compressed(os.path.join(HERE, 'arith_goldens.npz'), **allout)
    with open(os.path.join(HERE, 'arith_goldens_meta.json'), 'w') as f:
        json.dump(dict(
            generator='tests/golden/make_arith_goldens.py',
            source='reference classes cfl.models.dist.Dist / cfl.models.cfl.CFL (and everything they build) imported '
                   'from /root/reference and executed over tests/golden/tf_standin.py',
            note='TF primitives are float64 stand-ins restated from TensorFlow\'s documentation: the COMPOSITION of '
                 'the graph is the reference\'s, TensorFlow\'s own kernels were not run (TF is not installable here)',
            cases=metas), f, indent=1)
    print('wrote %d arrays' % len(allout))


if __name__ == '__main__':
    main()
