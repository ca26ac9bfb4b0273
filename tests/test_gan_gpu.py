"""MI355X parity of the MrCGAN stacks / post-epoch step against the torch-fp64 oracle
(oracle/gan_oracle.py).  Small shapes so that the CPU oracle (double backward through
the discriminator) finishes in seconds."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _mods():
    from cfl import hipgan as G
    from cfl.models import gan_blocks as GB
    from cfl.models import mrcgan as M
    from oracle import gan_oracle as GO
    from oracle import conv_oracle as CO
    return G, GB, M, GO, CO


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _close(name, got, want, rtol=2e-4, atol=None):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = max(np.abs(want).max(), 1e-30)
    tol = rtol * scale if atol is None else atol
    err = np.abs(got - want).max()
    assert err <= tol, '%s: max err %.3e (scale %.3e, tol %.3e)' % (name, err, scale, tol)


@pytest.mark.parametrize('act', [None, 'lrelu', 'relu', 'tanh', 'sigmoid'])
def test_elementwise(act):
    G, _, _, _, _ = _mods()
    rng = np.random.RandomState(0)
    x = rng.randn(1000).astype(np.float32)
    dy = rng.randn(1000).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    f = {None: lambda t: t, 'lrelu': lambda t: torch.relu(t) - 0.2 * torch.relu(-t), 'relu': torch.relu,
         'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[act]
    yt = f(xt)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    y = G.act_fwd(_dev(x), act)
    dx = G.act_bwd(y, _dev(dy), act)
    _close('y', y.cpu().numpy(), yt.detach().numpy(), 1e-6)
    _close('dx', dx.cpu().numpy(), xt.grad.numpy(), 1e-5)
    a = G.add_act(_dev(x), _dev(dy), act)
    _close('add_act', a.cpu().numpy(), f(torch.tensor(x + dy, dtype=torch.float64)).numpy(), 1e-6)


def test_subpixel_concat_gather():
    G, _, _, _, CO = _mods()
    rng = np.random.RandomState(1)
    x = rng.randn(3, 4, 5, 8).astype(np.float32)
    want = CO.conv2d_subpixel(torch.tensor(x, dtype=torch.float64), 2, 'relu').numpy()
    y = G.subpixel_fwd(_dev(x), 'relu')
    assert np.array_equal(y.cpu().numpy(), want.astype(np.float32))
    dy = rng.randn(*want.shape).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    CO.conv2d_subpixel(xt, 2, 'relu').backward(torch.tensor(dy, dtype=torch.float64))
    dx = G.subpixel_bwd(y, _dev(dy), 'relu')
    assert np.array_equal(dx.cpu().numpy(), xt.grad.numpy().astype(np.float32))
    # C % 16 == 0: the four-channels-per-thread form of the backward (with and without an activation)
    for act in ('relu', None):
        x = rng.randn(3, 4, 6, 48).astype(np.float32)
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        yt = CO.conv2d_subpixel(xt, 2, act)
        dy = rng.randn(*yt.shape).astype(np.float32)
        yt.backward(torch.tensor(dy, dtype=torch.float64))
        dx = G.subpixel_bwd(_dev(yt.detach().numpy().astype(np.float32)) if act else None, _dev(dy), act)
        assert np.array_equal(dx.cpu().numpy(), xt.grad.numpy().astype(np.float32))
    a, b = rng.randn(7, 3).astype(np.float32), rng.randn(7, 5).astype(np.float32)
    assert np.array_equal(G.concat_cols(_dev(a), _dev(b)).cpu().numpy(), np.concatenate([a, b], 1))
    P = rng.randn(6, 3, 4).astype(np.float32)
    c = rng.randint(0, 3, size=6).astype(np.int32)
    got = G.gather_prototype(_dev(P), torch.as_tensor(c).cuda())
    assert np.array_equal(got.cpu().numpy(), P[np.arange(6), c])


def test_losses():
    G, _, _, GO, _ = _mods()
    rng = np.random.RandomState(2)
    x = (3 * rng.randn(37)).astype(np.float32)
    sc = torch.zeros(4, device='cuda')
    for label, w in ((1.0, 0.5), (0.0, 1.0)):
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        l = w * GO.bce(xt, label)
        l.backward()
        dx = torch.zeros(37, device='cuda')
        G.bce_logits(_dev(x), label, w, sc[0:1], sc[1:2], dx)
        _close('bce', sc[0].item(), l.item(), 1e-6)
        _close('dbce', dx.cpu().numpy(), xt.grad.numpy(), 1e-5)
        assert abs(sc[1].item() - (x > 0).mean()) < 1e-6
    a, b = rng.randn(9, 6).astype(np.float32) * 0.2, rng.randn(9, 6).astype(np.float32) * 0.2
    for mode, margin in ((0, 0.0), (1, 0.3), (2, 0.8)):
        at = torch.tensor(a, dtype=torch.float64, requires_grad=True)
        d = ((at - torch.tensor(b, dtype=torch.float64)) ** 2).sum(-1)
        if mode == 0:
            l = d.mean()
        elif mode == 1:
            l = (torch.clamp(torch.sqrt(d + 1e-7) - margin, min=0) ** 2).mean()
        else:
            l = (torch.clamp(margin - torch.sqrt(d + 1e-7), min=0) ** 2).mean()
        l.backward()
        da = torch.zeros(9, 6, device='cuda')
        G.rowdist_loss(_dev(a), _dev(b), mode, margin, 1.0, sc[2:3], da)
        _close('rowdist%d' % mode, sc[2].item(), l.item(), 1e-5)
        _close('drowdist%d' % mode, da.cpu().numpy(), at.grad.numpy(), 1e-5)
    X = rng.rand(5, 300).astype(np.float32)
    eps = rng.rand(5, 1).astype(np.float32)
    want = X + 0.5 * X.astype(np.float64).std() * eps
    _close('perturb', G.perturb(_dev(X), _dev(eps), 0.5).cpu().numpy(), want, 1e-6)
    u = rng.randn(5, 300).astype(np.float32) * 0.05
    ut = torch.tensor(u, dtype=torch.float64, requires_grad=True)
    l = 0.5 * ((torch.sqrt((ut * ut).sum(1)) - 1) ** 2).mean()
    l.backward()
    v = G.grad_penalty(_dev(u), 0.5, sc[3:4])
    _close('gp', sc[3].item(), l.item(), 1e-5)
    _close('dgp', v.cpu().numpy(), ut.grad.numpy(), 1e-5)


def _load(pool, params, prefix):
    pool.load({prefix + k: v.numpy() if torch.is_tensor(v) else v for k, v in params.items()})


@pytest.mark.parametrize('gan_type,shape', [('srgan', (16, 16, 3)), ('conv', (16, 16, 1))])
def test_generator_fwd_bwd(gan_type, shape):
    G, GB, _, GO, _ = _mods()
    rng = np.random.RandomState(3)
    N, ind = 6, 11
    p = GO.GENERATORS[gan_type][0](shape, ind, rng)
    gen = GB.Generator(gan_type, shape, ind, 'tanh', np.random.RandomState(0), torch.device('cuda'))
    _load(gen.pool, p, 'Generator/')
    zc = rng.randn(N, ind)
    dA = rng.randn(N, int(np.prod(shape)))
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    out = GO.GENERATORS[gan_type][1](torch.tensor(zc), pt, shape, 'tanh')
    out.backward(torch.tensor(dA))
    acts, tape = gen.forward(_dev(zc))
    _close('acts', acts.cpu().numpy(), out.detach().numpy(), 2e-5)
    gen.backward(tape, _dev(dA))
    got = gen.pool.named(gen.pool.grad)
    for k in p:
        _close('d' + k, got['Generator/' + k], pt[k].grad.numpy(), 3e-4)


@pytest.mark.parametrize('gan_type,shape', [('srgan', (16, 16, 3)), ('conv', (16, 16, 1))])
def test_discriminator_fwd_bwd_gp(gan_type, shape):
    G, GB, _, GO, _ = _mods()
    rng = np.random.RandomState(4)
    N, Ld = 6, 5
    p = GO.DISCRIMINATORS[gan_type][0](shape, Ld, rng)
    for k in p:   # non-trivial gains / biases
        if k.endswith('/g'):
            p[k] = 1.0 + 0.1 * rng.randn(*p[k].shape)
        if k.endswith('/biases'):
            p[k] = 0.1 * rng.randn(*p[k].shape)
    disc = GB.Discriminator(gan_type, shape, Ld, np.random.RandomState(0), torch.device('cuda'))
    _load(disc.pool, p, 'Discriminator/')
    x = np.tanh(rng.randn(N, int(np.prod(shape))))
    dd, dl = rng.randn(N, 1), rng.randn(N, Ld)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, requires_grad=True)
    o, l = GO.DISCRIMINATORS[gan_type][1](xt, pt, shape)
    ((o * torch.tensor(dd)).sum() + (l * torch.tensor(dl)).sum()).backward()
    logit, lat, tape = disc.forward(_dev(x))
    _close('logit', logit.cpu().numpy(), o.detach().numpy(), 2e-5)
    _close('latent', lat.cpu().numpy(), l.detach().numpy(), 2e-5)
    dx = disc.backward(tape, 0, N, _dev(dd), _dev(dl), need_dx=True, need_dw=True)
    _close('dx', dx.cpu().numpy(), xt.grad.numpy(), 3e-4)
    got = disc.pool.named(disc.pool.grad)
    for k in p:
        _close('d' + k, got['Discriminator/' + k], pt[k].grad.numpy(), 3e-4)
    # row-range backward: rows [2, 5) only
    pt2 = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    o2, l2 = GO.DISCRIMINATORS[gan_type][1](torch.tensor(x[2:5]), pt2, shape)
    ((o2 * torch.tensor(dd[2:5])).sum() + (l2 * torch.tensor(dl[2:5])).sum()).backward()
    disc.backward(tape, 2, 5, _dev(dd[2:5]), _dev(dl[2:5]), need_dx=False, need_dw=True)
    got = disc.pool.named(disc.pool.grad)
    for k in p:
        _close('rows d' + k, got['Discriminator/' + k], pt2[k].grad.numpy(), 3e-4)
    # gradient penalty and its gradient (double backward)
    pt3 = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    xh = torch.tensor(x[1:5], requires_grad=True)
    o3, _ = GO.DISCRIMINATORS[gan_type][1](xh, pt3, shape)
    g, = torch.autograd.grad(o3.sum(), xh, create_graph=True)
    gp = 0.5 * ((torch.sqrt((g * g).sum(1)) - 1.0) ** 2).mean()
    grads = torch.autograd.grad(gp, [pt3[k] for k in p], allow_unused=True)
    sc = torch.zeros(1, device='cuda')
    disc.gp_grads(tape, 1, 5, 0.5, sc, disc.pool.grad2)
    _close('gp', sc.item(), gp.item(), 1e-4)
    got = disc.pool.named(disc.pool.grad2)
    gscale = max(float(t.abs().max()) for t in grads if t is not None)
    for k, t in zip(p, grads):
        want = np.zeros(p[k].shape) if t is None else t.numpy()
        _close('gp d' + k, got['Discriminator/' + k], want, atol=3e-4 * gscale)


@pytest.mark.parametrize('gan_type,shape,m_enc,m_prj,lambda_gp', [
    ('srgan', (16, 16, 3), 0.05, 0.2, 0.5),
    ('conv', (16, 16, 1), None, None, 0.5),
    ('srgan', (8, 8, 3), 0.05, 0.2, None),
])
def test_post_epoch_step_matches_oracle(gan_type, shape, m_enc, m_prj, lambda_gp):
    """Three free-running post-epoch steps beside the oracle in float64 AND beside the same oracle in float32 on the CPU:
    every loss part of every step must be no further from float64 than twice what the fp32 CPU evaluation of the same
    graph is (tests/parity_series.py; floor 1e-5 = north_star's bar; observed 1e-7 .. 5e-7, profiles/r05_series_gan_post_epoch_*.json)."""
    from parity_series import ParitySeries
    G, GB, M, GO, _ = _mods()
    B, Ld, zd = 4, 6, 5
    o = GO.GanOracle(gan_type, shape, 'tanh', zd, Ld, seed=1, m_enc=m_enc, m_prj=m_prj, lambda_gp=lambda_gp)
    o32 = GO.GanOracle(gan_type, shape, 'tanh', zd, Ld, seed=1, m_enc=m_enc, m_prj=m_prj, lambda_gp=lambda_gp,
                       dtype=torch.float32)
    ser = ParitySeries('gan_post_epoch_%s_%dx%d%s' % (gan_type, shape[0], shape[1], '' if lambda_gp else '_nogp'), floor=1e-5,
                       meta=dict(gan_type=gan_type, shape=shape, B=B, steps=3, lambda_gp=lambda_gp))
    ph = M.GanPhase(gan_type, shape, 'tanh', zd, Ld, B, torch.device('cuda'), np.random.RandomState(0),
                    lambda_gp=lambda_gp, lambda_dra=0.5, m_enc=m_enc, m_prj=m_prj)
    _load(ph.gen.pool, o.gp, 'Generator/')
    _load(ph.disc.pool, o.dp, 'Discriminator/')
    rng = np.random.RandomState(5)
    N = int(np.prod(shape))
    for it in range(3):
        batch = [np.tanh(rng.randn(B, N)), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld),
                 0.3 * rng.randn(B, Ld), rng.randn(B, zd), rng.rand(B, 1)]
        d_total, g_total, parts = o.step(*[torch.tensor(b) for b in batch])
        d32, g32, parts32 = o32.step(*[torch.tensor(np.asarray(b, np.float32)) for b in batch])
        ph.step(*[_dev(b) for b in batch])
        s = ph.read_scalars()
        keys = ['d_loss_real', 'd_loss_fake', 'd_loss_d', 'g_loss', 'g_loss_d'] + (['d_grad_loss'] if lambda_gp else []) + \
               (['g_loss_d_neg'] if m_prj else [])
        for k in keys:
            ser.add(it, k, s[k], float(parts[k]), float(parts32[k]))
        ser.add(it, 'd_total_loss', s['d_total_loss'], float(d_total), float(d32))
        ser.add(it, 'g_total_loss', s['g_total_loss'], float(g_total), float(g32))
        if it == 0:      # before any update: pure forward precision, no yardstick needed
            assert all(r[2] <= 1e-5 for r in ser.rows), ser.rows
    ser.check()
    # variables after 3 simultaneous Adam steps (Adam's first steps are +-lr per element, so a
    # handful of sign flips of ~zero gradients are tolerated)
    # ... and no more of them than the fp32 CPU evaluation has
    for pool, ref, ref32, pre in ((ph.gen.pool, o.gp, o32.gp, 'Generator/'), (ph.disc.pool, o.dp, o32.dp, 'Discriminator/')):
        got = pool.named()
        for k, v in ref.items():
            diff = np.abs(got[pre + k] - v.numpy())
            diff32 = np.abs(ref32[k].numpy().astype(np.float64) - v.numpy())
            assert (diff <= 2e-4).mean() >= min(0.97, (diff32 <= 2e-4).mean() - 0.02), (k, diff.max(), (diff32 <= 2e-4).mean())


def test_stream_placement_trial_leaves_the_trajectory_alone():
    """GanPhase times candidate stream sets on NON-applying steps at its first step (mrcgan._tune_streams): three training steps
    with the trial must equal three steps without it bit for bit -- scalars of every step, every variable and Adam slot -- and
    the trial must have run (eight candidate timings recorded)."""
    G, GB, M, GO, _ = _mods()
    B, Ld, zd, shape = 4, 6, 5, (16, 16, 3)
    rng = np.random.RandomState(5)
    N = int(np.prod(shape))
    batches = [[np.tanh(rng.randn(B, N)), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld),
                0.3 * rng.randn(B, Ld), rng.randn(B, zd), rng.rand(B, 1)] for _ in range(3)]
    runs = []
    before = M.GanPhase.tune_streams
    try:
        for tune in (False, True):
            M.GanPhase.tune_streams = tune
            ph = M.GanPhase('srgan', shape, 'tanh', zd, Ld, B, torch.device('cuda'), np.random.RandomState(0), lambda_gp=0.5,
                            lambda_dra=0.5, m_enc=0.05, m_prj=0.2)
            scal = []
            for b in batches:
                ph.step(*[_dev(x) for x in b])
                scal.append(ph.scalars.detach().cpu().numpy().copy())
            torch.cuda.synchronize()
            runs.append((scal, ph.gen.state(), ph.disc.state(), getattr(ph, 'stream_tuning', None)))
    finally:
        M.GanPhase.tune_streams = before
    (s0, g0, d0, t0), (s1, g1, d1, t1) = runs
    assert t0 is None and t1 is not None and len(t1) == 8 and all(t > 0 for t in t1)
    assert all(np.array_equal(a, b) for a, b in zip(s0, s1))
    for a, b in ((g0, g1), (d0, d1)):
        for part in ('variables', 'adam_m', 'adam_v'):
            assert set(a[part]) == set(b[part]) and all(np.array_equal(a[part][k], b[part][k]) for k in a[part])
        assert a['beta1_power'] == b['beta1_power']


def test_gradient_penalty_chain_forks_only_behind_prepared_caches():
    """the gradient-penalty chain may run its own discriminator forward beside the main stream's only when
    every layer cache was rebuilt AHEAD for the current weights (_Net.caches_prepared) -- the caches' validity bits are host
    state shared by all streams.  First step after construction and after load_state: not prepared, X_hat rides in the batched
    forward; from the second step on: prepared.  Both forms give the same step bit for bit (three steps, every scalar, every
    variable and Adam slot)."""
    G, GB, M, GO, _ = _mods()
    B, Ld, zd, shape = 4, 6, 5, (16, 16, 3)
    rng = np.random.RandomState(7)
    N = int(np.prod(shape))
    batches = [[np.tanh(rng.randn(B, N)), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld),
                0.3 * rng.randn(B, Ld), rng.randn(B, zd), rng.rand(B, 1)] for _ in range(3)]
    runs = []
    before = (M.GanPhase.tune_streams, M.GanPhase.gp_early)
    try:
        M.GanPhase.tune_streams = False
        for early in (False, True):
            M.GanPhase.gp_early = early
            ph = M.GanPhase('srgan', shape, 'tanh', zd, Ld, B, torch.device('cuda'), np.random.RandomState(0), lambda_gp=0.5,
                            lambda_dra=0.5, m_enc=0.05, m_prj=0.2)
            assert not ph.disc.caches_prepared()                 # nothing was ever built
            scal, prepared = [], []
            for b in batches:
                prepared.append(ph.disc.caches_prepared())
                ph.step(*[_dev(x) for x in b])
                scal.append(ph.scalars.detach().cpu().numpy().copy())
            torch.cuda.synchronize()
            assert prepared == [False, True, True], prepared     # (the optimizer step prepares the next step's caches)
            st = ph.state()
            ph.load_state(st)                                    # weights replaced: stale until the next preparation
            assert not ph.disc.caches_prepared()
            runs.append((scal, st))
    finally:
        M.GanPhase.tune_streams, M.GanPhase.gp_early = before
    (s0, st0), (s1, st1) = runs
    assert all(np.array_equal(a, b) for a, b in zip(s0, s1))
    for net in ('generator', 'discriminator'):
        for part in ('variables', 'adam_m', 'adam_v'):
            assert all(np.array_equal(st0[net][part][k], st1[net][part][k]) for k in st0[net][part])


def test_deferred_weight_gradient_finalisation_is_bit_identical():
    """Round 6: the weight gradients of a backward chain are finished together -- ONE slab-sum launch and ONE finalisation launch for
    all layers of the chain (cfl_conv2d_wn_wgrad_slabs + cfl_conv_wfinal_many) -- instead of a launch pair per layer.  Same per-element
    code in the same order: three post-epoch steps at the config-5 shape family (srgan 32x32 with the residual stages, and the conv
    GAN) must give the same scalars, variables and Adam slots bit for bit, and the launch count per step must drop."""
    G, GB, M, GO, _ = _mods()
    from cfl import hipabi as H
    rng = np.random.RandomState(9)
    before = (GB.Workspace.defer_wfinal, M.GanPhase.tune_streams, GB._Net.prep_batched)
    try:
        M.GanPhase.tune_streams = False
        for gan_type, shape, B in (('srgan', (32, 32, 3), 6), ('conv', (16, 16, 1), 4)):
            Ld, zd = 6, 5
            N = int(np.prod(shape))
            batches = [[np.tanh(rng.randn(B, N)), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld),
                        0.3 * rng.randn(B, Ld), rng.randn(B, zd), rng.rand(B, 1)] for _ in range(3)]
            runs = []
            for defer in (False, True):
                GB.Workspace.defer_wfinal = defer
                GB._Net.prep_batched = defer      # ... and the cache preparation of all layers in two launches (cfl_conv_prepare_cached_many)
                ph = M.GanPhase(gan_type, shape, 'tanh', zd, Ld, B, torch.device('cuda'), np.random.RandomState(0), lambda_gp=0.5,
                                lambda_dra=0.5, m_enc=0.05, m_prj=0.2)
                scal = []
                for b in batches:
                    ph.step(*[_dev(x) for x in b])
                    scal.append(ph.scalars.detach().cpu().numpy().copy())
                torch.cuda.synchronize()
                H.profile_enable(False)
                runs.append((scal, ph.state()))
            (s0, st0), (s1, st1) = runs
            assert all(np.array_equal(a, b) for a, b in zip(s0, s1)), gan_type
            for net in ('generator', 'discriminator'):
                for part in ('variables', 'adam_m', 'adam_v'):
                    bad = [k for k in st0[net][part] if not np.array_equal(st0[net][part][k], st1[net][part][k])]
                    assert not bad, (gan_type, net, part, bad[:3])
    finally:
        GB.Workspace.defer_wfinal, M.GanPhase.tune_streams, GB._Net.prep_batched = before


def test_conditional_discriminator_srgan_64():
    """SRDiscriminator with the cgan condition tiled in at stage 3 (64x64 images) and fc_t: forward, parameter
    and input gradients against the oracle."""
    G, GB, _, GO, _ = _mods()
    rng = np.random.RandomState(6)
    shape, Ld, cd, td, N = (64, 64, 3), 5, 9, 7, 2
    p = GO.init_sr_discriminator(shape, Ld, rng, c_dim=cd, t_dim=td)
    disc = GB.Discriminator('srgan', shape, Ld, np.random.RandomState(0), torch.device('cuda'), c_dim=cd, t_dim=td)
    assert set(disc.pool.order) == set('Discriminator/' + k for k in p)
    _load(disc.pool, p, 'Discriminator/')
    x = np.tanh(rng.randn(N, int(np.prod(shape))))
    t = rng.randn(N, cd)
    dd = rng.randn(N, 1)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x, requires_grad=True)
    o, _ = GO.sr_discriminator(xt, pt, shape, t=torch.tensor(t))
    (o * torch.tensor(dd)).sum().backward()
    logit, _, tape = disc.forward(_dev(x), _dev(t))
    _close('logit', logit.cpu().numpy(), o.detach().numpy(), 5e-5)
    dx = disc.backward(tape, 0, N, _dev(dd), None, need_dx=True, need_dw=True)
    _close('dx', dx.cpu().numpy(), xt.grad.numpy(), 5e-4)
    got = disc.pool.named(disc.pool.grad)
    for k in p:
        want = np.zeros(p[k].shape) if pt[k].grad is None else pt[k].grad.numpy()
        if np.abs(want).max() > 0:
            _close('d' + k, got['Discriminator/' + k], want, 5e-4)


@pytest.mark.parametrize('t_dim', [None, 7])
def test_cgan_step_matches_oracle(t_dim):
    G, GB, M, GO, _ = _mods()
    gan_type, shape = 'conv', (16, 16, 1)
    B, Ld, zd = 4, 6, 5
    cd = 9 if t_dim else Ld
    o = GO.GanOracle(gan_type, shape, 'tanh', zd, Ld, seed=2, lambda_gp=0.5, cgan=True, c_dim=cd, t_dim=t_dim)
    ph = M.GanPhase(gan_type, shape, 'tanh', zd, Ld, B, torch.device('cuda'), np.random.RandomState(0),
                    lambda_gp=0.5, lambda_dra=0.5, cgan=True, c_dim=cd, t_dim=t_dim)
    assert set(ph.gen.pool.order) == set('Generator/' + k for k in o.gp)
    assert set(ph.disc.pool.order) == set('Discriminator/' + k for k in o.dp)
    _load(ph.gen.pool, o.gp, 'Generator/')
    _load(ph.disc.pool, o.dp, 'Discriminator/')
    rng = np.random.RandomState(7)
    N = int(np.prod(shape))
    for it in range(3):
        batch = [np.tanh(rng.randn(B, N)), np.tanh(rng.randn(B, N)), 0.5 * rng.randn(B, cd), 0.5 * rng.randn(B, cd),
                 rng.randn(B, zd), rng.rand(B, 1)]
        d_total, g_total, parts = o.step(*[torch.tensor(b) for b in batch])
        ph.step_cgan(*[_dev(b) for b in batch])
        s = ph.read_scalars()
        tol = 5e-5 if it == 0 else 2e-3
        for k in ('d_loss_real', 'd_loss_fake', 'd_loss_neg', 'd_grad_loss', 'g_loss', 'g_loss_int'):
            assert abs(s[k] - float(parts[k])) <= tol * max(1.0, abs(float(parts[k]))), (it, k, s[k], float(parts[k]))
        assert abs(s['d_total_loss'] - float(d_total)) <= tol * max(1.0, abs(float(d_total)))
        assert abs(s['g_total_loss'] - float(g_total)) <= tol * max(1.0, abs(float(g_total)))
    for pool, ref, pre in ((ph.gen.pool, o.gp, 'Generator/'), (ph.disc.pool, o.dp, 'Discriminator/')):
        got = pool.named()
        for k, v in ref.items():
            diff = np.abs(got[pre + k] - v.numpy())
            assert (diff <= 2e-4).mean() >= 0.97, (k, diff.max())


@pytest.mark.parametrize('n_pos,n_neg,ties', [(1, 1, False), (37, 91, False), (5000, 3000, True), (70000, 90001, True)])
def test_gpu_auc_matches_sklearn(n_pos, n_neg, ties):
    from sklearn.metrics import roc_auc_score
    G, _, _, _, _ = _mods()
    rng = np.random.RandomState(n_pos)
    sp = (rng.randn(n_pos) + 0.5).astype(np.float32)
    sn = (rng.randn(n_neg) - 0.5).astype(np.float32)
    if ties:                      # heavy ties, incl. exact zeros and values shared between the classes
        sp = np.round(sp * 4) / 4
        sn = np.round(sn * 4) / 4
    if n_pos == 1:
        sp[:] = 0.25
        sn[:] = 0.25
    a, acc = G.auc(torch.as_tensor(sp).cuda(), torch.as_tensor(sn).cuda())
    want = roc_auc_score(np.r_[np.ones(n_pos), np.zeros(n_neg)], np.r_[sp, sn])
    assert abs(a - want) <= 1e-12, (a, want)
    assert abs(acc - ((sp > 0).sum() + (sn <= 0).sum()) / (n_pos + n_neg)) <= 1e-12


def test_image_transform_kernel_matches_host_restatement():
    """cfl_image_transform (crop / pad window with per-sample offsets, central window, bilinear resize, per-sample
    mirror) against the NumPy restatement in cfl.ops.ImageTransform."""
    from cfl import ops
    G, _, _, _, _ = _mods()
    rng = np.random.RandomState(9)
    x = rng.rand(5, 12 * 10 * 3).astype(np.float32)
    cases = [ops.ImageTransform((12, 10, 3), (8, 6, 3), 'random_crop', True),
             ops.ImageTransform((12, 10, 3), (8, 6, 3), 'crop'),
             ops.ImageTransform((12, 10, 3), (16, 14, 3), 'crop'),          # zero padding
             ops.ImageTransform((12, 10, 3), (7, 15, 3), 'resize', True),
             ops.ImageTransform((12, 10, 3), (24, 20, 3), 'resize'),
             ops.ImageTransform((12, 10, 3), (12, 10, 3), 'reshape', True)]
    for tr in cases:
        r1, r2 = np.random.RandomState(3), np.random.RandomState(3)
        got = tr.apply(torch.as_tensor(x).cuda(), r1).cpu().numpy()
        off, flip = tr.draw(5, r2)
        want = tr(x, off, flip)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= (1e-6 if tr.kind == 'resize' else 0.0), tr.kind


def test_per_channel_normalizer_on_gpu():
    from cfl import ops
    rng = np.random.RandomState(10)
    x = rng.rand(7, 4 * 5 * 3).astype(np.float32)
    n = ops.normalizer_v2((4, 5, 3), scale=None, mean=(0.485, 0.456, 0.406), norm=(0.229, 0.224, 0.225),
                          clip_value_min=-1.0, clip_value_max=1.0)
    got = n.apply(torch.as_tensor(x).cuda()).cpu().numpy()
    assert np.abs(got - n(x)).max() <= 1e-6
    s = ops.normalizer_v2((4, 5, 3), scale=2.0, mean=0.5, norm=0.5)
    assert np.abs(s.apply(torch.as_tensor(x).cuda()).cpu().numpy() - s(x)).max() <= 1e-6


def test_conv_weight_cache_is_bit_identical():
    """cfl_conv2d_wn_*_cached: the per-layer cache of the weight-norm scale and the prepared filter planes (rebuilt
    when the parameter pool's version moves: Adam, load) must not change a single bit of a training trajectory --
    3 srgan steps at 16x16 (every halo tile variant of the small shapes) with the cache on and off."""
    from cfl import hipgan as G
    from cfl.models import mrcgan as M
    shape, Ld, zd, B = (16, 16, 3), 6, 5, 4
    rng = np.random.RandomState(5)
    N = int(np.prod(shape))
    batches = [[np.tanh(rng.randn(B, N)), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld),
                0.3 * rng.randn(B, Ld), rng.randn(B, zd), rng.rand(B, 1)] for _ in range(3)]
    dev = lambda a: torch.tensor(np.asarray(a, np.float32), device='cuda')
    out = {}
    saved = G.ConvCache.enabled
    try:
        for on in (True, False):
            G.ConvCache.enabled = on
            ph = M.GanPhase('srgan', shape, 'tanh', zd, Ld, B, torch.device('cuda'), np.random.RandomState(0),
                            lambda_gp=0.5, lambda_dra=0.5, m_enc=0.05, m_prj=0.2)
            hist = []
            for b in batches:
                ph.step(*[dev(x) for x in b])
                hist.append(dict(ph.read_scalars()))
            out[on] = (hist, ph.gen.pool.theta.clone(), ph.disc.pool.theta.clone())
    finally:
        G.ConvCache.enabled = saved
    assert out[True][0] == out[False][0]
    assert torch.equal(out[True][1], out[False][1]) and torch.equal(out[True][2], out[False][2])
