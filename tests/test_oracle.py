"""Pins for the arithmetic oracle (oracle/cfl_oracle.py).

The reference has no tests or golden vectors for its TF arithmetic,
so the oracle is pinned by: analytic known-answer tests,
float64 finite differences of every analytic gradient, and an independent
torch-autograd float64 restatement of the same forward.
"""
import itertools
import os

import numpy as np
import pytest
import torch

from oracle import cfl_oracle as O


def test_kat_pcd():
    v = np.array([[0.0, 0.0]])
    P = np.array([[[1.0, 0.0], [0.0, 2.0]]])
    d, (s, m) = O.dist_pcd(v, P)
    assert np.allclose(s[0], [0.95257413, 0.04742587], atol=1e-8)
    assert np.allclose(m[0], [0.95257413, 0.09485175], atol=1e-8)
    assert abs(d[0] - 0.91639432087814) < 1e-12
    o = 1e-6 - d[0]
    assert abs(o - (-0.91639332087814)) < 1e-12
    assert abs(O.bce_with_logits(o, 1.0) - 1.2528362474292705) < 1e-12
    assert abs(O.bce_with_logits(o, 0.0) - 0.33644292655113045) < 1e-12


def test_kat_monomer():
    a = np.array([[0.0, 0.0]])
    P = np.array([[[1.0, 0.0], [0.0, 2.0]]])
    u = np.array([[0.0, np.log(3.0)]])
    d, (w, e) = O.dist_monomer(a, u, P)
    assert np.allclose(w[0], [0.25, 0.75])
    assert np.allclose(e[0], [1.0, 4.0])
    assert abs(d[0] - 3.25) < 1e-12


def test_kat_pcd_k1_and_siamese():
    v = np.array([[1.0, 2.0, 3.0]])
    P = np.array([[[0.0, 0.0, 1.0]]])
    d, _ = O.dist_pcd(v, P)
    assert d[0] == 1 + 4 + 4
    assert O.dist_siamese(v, P[:, 0])[0] == 9.0


def test_threshold_floor_and_mask():
    cfg = O.EncoderCfg(D=8, L=3, K=2)
    rng = np.random.RandomState(1)
    p = O.init_encoder_params(cfg, rng, np.float64)
    x = rng.randn(4, 8)
    s_lo = O.pair_scores(cfg, p, np.float64(-5.0), x, x[::-1])
    s_fl = O.pair_scores(cfg, p, np.float64(1e-6), x, x[::-1])
    assert np.array_equal(s_lo, s_fl)          # max(thr, 1e-6)


def _flatten(params):
    keys = sorted(params)
    return keys, np.concatenate([params[k].ravel() for k in keys])


def _unflatten(keys, like, vec):
    out, o = {}, 0
    for k in keys:
        n = like[k].size
        out[k] = vec[o:o + n].reshape(like[k].shape)
        o += n
    return out


CASES = []
for style, dist_type, K, act in itertools.product(
        ('dist', 'cfl'), ('pcd', 'monomer', 'siamese'), (1, 3),
        (None, 'tanh')):
    if style == 'dist' and (dist_type != 'pcd' or act):
        continue
    if dist_type == 'siamese' and K != 1:
        continue
    CASES.append((style, dist_type, K, act))


@pytest.mark.parametrize('style,dist_type,K,act', CASES)
@pytest.mark.parametrize('loss', ['thr', 'thr_pw_reg', 'caffe', 'lambda_m',
                                  'no_ut'])
def test_gradients_finite_difference(style, dist_type, K, act, loss):
    if loss == 'caffe' and dist_type != 'siamese':
        pytest.skip('caffe margin is siamese-only (cfl/utils.py:66-67)')
    rng = np.random.RandomState(7)
    cfg = O.EncoderCfg(D=6, L=4, K=K, dist_type=dist_type, style=style,
                       act_type=act)
    lcfg = dict(
        thr=O.LossCfg(),
        thr_pw_reg=O.LossCfg(pos_weight=0.25, reg_const=5e-2),
        caffe=O.LossCfg(use_threshold=False, caffe_margin=3.0, pos_weight=0.5),
        lambda_m=O.LossCfg(lambda_m=0.5, pos_weight=0.0625),
        no_ut=O.LossCfg(use_threshold=False, lambda_m=0.3),
    )[loss]
    p = O.init_encoder_params(cfg, rng, np.float64)
    for k in p:                       # move biases/g off their init values
        p[k] = p[k] + 0.1 * rng.randn(*p[k].shape)
    B = 5
    batch = tuple(rng.randn(B, cfg.D) for _ in range(4))
    thr = np.float64(0.7)
    sc, g, gd, dthr, dthr_aux = O.train_step_loss_and_grads(
        cfg, lcfg, p, thr, batch)
    keys, vec = _flatten(p)

    def f(vv, t=thr, which='total'):
        pp = _unflatten(keys, p, vv)
        return O.train_step_loss_and_grads(cfg, lcfg, pp, t, batch)[0][which]

    eps = 1e-6
    num = np.zeros_like(vec)
    for i in range(vec.size):
        e = np.zeros_like(vec)
        e[i] = eps
        num[i] = (f(vec + e) - f(vec - e)) / (2 * eps)
    ana = np.concatenate([np.asarray(g.get(k, np.zeros_like(p[k]))).ravel()
                          for k in keys])
    assert np.abs(num - ana).max() < 5e-8, np.abs(num - ana).max()
    nthr = (f(vec, thr + eps) - f(vec, thr - eps)) / (2 * eps)
    assert abs(nthr - dthr) < 5e-8
    nthr_aux = (f(vec, thr + eps, 'thres') - f(vec, thr - eps, 'thres')) / (2 * eps)
    assert abs(nthr_aux - dthr_aux) < 5e-8


def test_gradients_directed_fd():
    rng = np.random.RandomState(3)
    cfg = O.EncoderCfg(D=5, L=3, K=2, dist_type='pcd', style='cfl')
    lcfg = O.LossCfg(reg_const=1e-2, lambda_m=0.5)
    ps = O.init_encoder_params(cfg, rng, np.float64)
    pd = O.init_encoder_params(cfg, rng, np.float64)
    batch = tuple(rng.randn(4, cfg.D) for _ in range(4))
    thr = np.float64(0.3)
    sc, g, gd, dthr, _ = O.train_step_loss_and_grads(cfg, lcfg, ps, thr, batch, pd)
    eps = 1e-6
    for params, grads, is_dst in ((ps, g, False), (pd, gd, True)):
        for k in params:
            it = np.nditer(params[k], flags=['multi_index'])
            for _ in it:
                idx = it.multi_index
                old = params[k][idx]
                params[k][idx] = old + eps
                fp = O.train_step_loss_and_grads(cfg, lcfg, ps, thr, batch, pd)[0]['total']
                params[k][idx] = old - eps
                fm = O.train_step_loss_and_grads(cfg, lcfg, ps, thr, batch, pd)[0]['total']
                params[k][idx] = old
                a = grads.get(k, np.zeros_like(params[k]))[idx]
                assert abs((fp - fm) / (2 * eps) - a) < 5e-8, (k, idx)


def _torch_forward(cfg, lcfg, p, thr, batch):
    """Independent restatement with torch ops + autograd (float64)."""
    def head(x, name):
        W = p[name + '/W']
        y = x @ W
        if cfg.weight_norm:
            y = y * (p[name + '/g'] / torch.sqrt((W * W).sum(0)))
        if name + '/b' in p:
            y = y + p[name + '/b']
        return y

    def act(y):
        return {None: lambda t: t, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid,
                'relu': torch.relu}[cfg.act_type](y)

    def dist(xs, xt):
        B = xs.shape[0]
        if cfg.dist_type == 'pcd':
            P = act(head(xs, 'proto')).reshape(B, cfg.K, cfg.L)
            v = act(head(xt, 'outputs'))
            if cfg.K > 1:
                logits = -((v[:, None, :] - P) ** 2).sum(-1)
                s = torch.softmax(logits, -1)
                m = (P * s[:, :, None]).sum(-2)
                return ((v - m) ** 2).sum(-1)
            return ((v - P[:, 0]) ** 2).sum(-1)
        if cfg.dist_type == 'monomer':
            ya = head(xs, 'outputs')
            u = head(ya, 'mono')
            P = act(head(xt, 'proto')).reshape(B, cfg.K, cfg.L)
            e = ((act(ya)[:, None, :] - P) ** 2).sum(-1)
            return (torch.softmax(u, -1) * e).sum(-1)
        return ((act(head(xs, 'outputs')) - act(head(xt, 'outputs'))) ** 2).sum(-1)

    xps, xpd, xns, xnd = batch
    dp, dn = dist(xps, xpd), dist(xns, xnd)
    t = torch.maximum(thr, torch.tensor(1e-6, dtype=thr.dtype))
    op, on = t - dp, t - dn
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    lp = bce(op, torch.ones_like(op))
    ln = bce(on, torch.zeros_like(on))
    pw = lcfg.pos_weight if lcfg.pos_weight else 1.0
    total = sum(lcfg.reg_const * 0.5 * (w * w).sum() for k, w in p.items()
                if not k.endswith('/g')) if lcfg.reg_const else 0.0
    if lcfg.use_threshold:
        total = total + lp * pw + ln
    if lcfg.caffe_margin:
        total = total + 0.5 * (dp.mean() * pw + torch.clamp(lcfg.caffe_margin - dn, min=0).mean())
    elif lcfg.lambda_m:
        total = total + dp.mean() * lcfg.lambda_m * pw
    return total, lp, ln


@pytest.mark.parametrize('style,dist_type,K,act', CASES)
def test_against_torch_autograd(style, dist_type, K, act):
    rng = np.random.RandomState(11)
    cfg = O.EncoderCfg(D=16, L=5, K=K, dist_type=dist_type, style=style, act_type=act)
    lcfg = O.LossCfg(pos_weight=0.25, reg_const=1e-3, lambda_m=0.5)
    p = O.init_encoder_params(cfg, rng, np.float64)
    for k in p:
        p[k] = p[k] + 0.05 * rng.randn(*p[k].shape)
    batch = tuple(rng.randn(9, cfg.D) for _ in range(4))
    thr = np.float64(0.4)
    sc, g, _, dthr, _ = O.train_step_loss_and_grads(cfg, lcfg, p, thr, batch)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    tthr = torch.tensor(thr, requires_grad=True)
    total, lp, ln = _torch_forward(cfg, lcfg, tp, tthr, tuple(torch.tensor(b) for b in batch))
    total.backward()
    assert abs(total.item() - sc['total']) < 1e-12
    assert abs(lp.item() - sc['loss_pos']) < 1e-12
    assert abs(ln.item() - sc['loss_neg']) < 1e-12
    for k in p:
        ref = tp[k].grad.numpy() if tp[k].grad is not None else np.zeros_like(p[k])
        assert np.abs(ref - g.get(k, 0)).max() < 1e-12, k
    assert abs(tthr.grad.item() - dthr) < 1e-12


def test_adam_tf_semantics():
    """TF-1.x Adam: eps outside the sqrt, lr_t from the power accumulators
     -- differs from torch.optim.Adam's eps placement."""
    rng = np.random.RandomState(0)
    th = {'w': rng.randn(7).astype(np.float64)}
    st = O.AdamState(lr=1e-2)
    ref = th['w'].copy()
    m = np.zeros(7)
    v = np.zeros(7)
    for t in range(1, 6):
        g = rng.randn(7)
        st.apply(th, {'w': g})
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        lr_t = 1e-2 * np.sqrt(1 - np.float64(np.float32(0.999)) ** t) / (1 - np.float64(np.float32(0.9)) ** t)
        ref = ref - lr_t * m / (np.sqrt(v) + 1e-8)
        assert np.allclose(th['w'], ref, rtol=1e-6, atol=1e-9)


def test_oracle_trainer_learns():
    rng = np.random.RandomState(5)
    cfg = O.EncoderCfg(D=32, L=4, K=2)
    tr = O.OracleTrainer(cfg, O.LossCfg(), lr=1e-2, dtype=np.float64)
    A = rng.randn(32, 32) * 0.3
    xs = rng.randn(256, 32)
    pos = (xs, xs @ A + 0.05 * rng.randn(256, 32))
    neg = (rng.randn(256, 32), rng.randn(256, 32))
    first = last = None
    for i in range(60):
        sc = tr.step((pos[0], pos[1], neg[0], neg[1]))
        first = first if first is not None else sc['total']
        last = sc['total']
    assert last < first
    ev = O.dist_eval(tr.scores(*pos), tr.scores(*neg))
    assert ev['auc'] > 0.8


def test_loader_oracle_reads_what_the_product_reader_reads(tmp_path):
    """oracle/loader_oracle.py (the reference's per-row seek + read, cfl/input_data.py:212-228; bench.py's cpu_baseline.loader leg)
    against the product's reader of the same file -- itself pinned to goldens captured from the reference's own module
    (tests/test_input_data.py): same bytes, and the file it writes has the reference's record grammar."""
    from cfl import input_data
    from oracle import loader_oracle as LO
    rng = np.random.RandomState(3)
    path = str(tmp_path / 'features.b')
    D, n = 24, 17
    LO.write_features(path, rng, n, D)
    assert os.path.getsize(path) == n * (10 + 4 * D)
    pos = np.array([[0, 16], [5, 5], [16, 1], [3, 9]])
    got = LO.labeled_batch_by_seek(path, pos, pos[::-1], D)
    for g, want_pos in zip(got, (pos[:, 0], pos[:, 1], pos[::-1][:, 0], pos[::-1][:, 1])):
        want = input_data.load_features_by_positions(path, want_pos, D)
        assert g.dtype == np.float32 and np.array_equal(g, want)
    assert input_data.load_asins_by_positions(path, [0, 16], D) == ['0000000000', '0000000016']
