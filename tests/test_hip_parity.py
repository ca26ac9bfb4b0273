"""GPU parity: the HIP path (through the C ABI) against the NumPy oracle on the
same seeded inputs.  Tolerances: the north star asks loss / score parity within
1e-5 in fp32; gradients are checked relative to the largest gradient entry.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cfl_oracle as O

pytestmark = pytest.mark.gpu

H = None


@pytest.fixture(scope='module', autouse=True)
def _hip():
    global H
    from cfl import hipabi
    hipabi.lib()                      # raises if the extension is missing
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    H = hipabi
    yield


def _mk(cfg, rng, perturb=0.05):
    p = O.init_encoder_params(cfg, rng, np.float32)
    for k in p:
        p[k] = (p[k] + perturb * rng.randn(*p[k].shape).astype(np.float32) *
                (0.1 if k.endswith('/W') else 1.0)).astype(np.float32)
    return p


def _shape(cfg, directed=False):
    return H.make_shape(cfg.D, cfg.L, cfg.K, cfg.dist_type, cfg.weight_norm,
                        cfg.has_bias, cfg.act_type, directed)


def _inputs(rng, n, D, scale):
    return (np.abs(rng.randn(n, D)) * scale).astype(np.float32)


def _to64(p):
    return None if p is None else {k: v.astype(np.float64) for k, v in p.items()}


SCORE_CASES = [
    # style, dist, D, L, K, act, n, norm
    ('dist', 'pcd', 4096, 20, 3, None, 500, 58.388599),
    ('dist', 'pcd', 4096, 10, 4, None, 100, 58.388599),
    ('dist', 'pcd', 2048, 20, 5, None, 333, 1.0),
    ('dist', 'pcd', 256, 7, 1, None, 17, 1.0),
    ('cfl', 'pcd', 1024, 64, 3, None, 250, 31.9098),
    ('cfl', 'monomer', 1024, 64, 3, None, 250, 31.9098),
    ('cfl', 'siamese', 1024, 256, 1, None, 129, 31.9098),
    ('cfl', 'pcd', 512, 20, 4, 'tanh', 64, 1.0),
    ('cfl', 'monomer', 512, 12, 2, 'sigmoid', 64, 1.0),
]


@pytest.mark.parametrize('style,dist,D,L,K,act,n,nv', SCORE_CASES)
def test_pair_scores(style, dist, D, L, K, act, n, nv):
    rng = np.random.RandomState(1234)
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style=style, act_type=act)
    p = _mk(cfg, rng)
    thr = 0.37
    xs = _inputs(rng, n, D, nv / 4)
    xt = _inputs(rng, n, D, nv / 4)
    ref = O.pair_scores(cfg, _to64(p), np.float64(thr),
                        xs.astype(np.float64) / nv, xt.astype(np.float64) / nv)
    sh = _shape(cfg)
    theta = H.pack_theta(sh, p, None, thr, 'cuda')
    ws = torch.empty(H.workspace_bytes(sh, n, 1) // 4, dtype=torch.float32, device='cuda')
    dists = torch.empty(n, device='cuda')
    got = H.pair_scores(sh, H.make_norm(1.0 / nv), torch.from_numpy(xs).cuda(),
                        torch.from_numpy(xt).cuda(), theta, ws, dists=dists)
    got = got.cpu().numpy().astype(np.float64)
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(got - ref).max() <= 1e-5 * scale, (np.abs(got - ref).max(), scale)
    assert np.abs((thr - dists.cpu().numpy()) - got).max() <= 1e-6 * scale


def test_pair_scores_elementwise_norm():
    """clip + mean path of cfl/ops.py:66-124 (not foldable into the epilogue)."""
    rng = np.random.RandomState(5)
    D, n = 256, 50
    cfg = O.EncoderCfg(D=D, L=8, K=2, dist_type='pcd', style='cfl')
    p = _mk(cfg, rng)
    xs = rng.randn(n, D).astype(np.float32)
    xt = rng.randn(n, D).astype(np.float32)
    f = lambda x: O.normalize_v2(x.astype(np.float64), scale=None, mean=0.5, norm=0.5,
                                 clip_min=-1.0, clip_max=1.0)
    ref = O.pair_scores(cfg, _to64(p), np.float64(1e-6), f(xs), f(xt))
    sh = _shape(cfg)
    theta = H.pack_theta(sh, p, None, 1e-6, 'cuda')
    ws = torch.empty(H.workspace_bytes(sh, n, 1) // 4, dtype=torch.float32, device='cuda')
    got = H.pair_scores(sh, H.make_norm(1 / 0.5, -0.5 / 0.5, -1.0, 1.0),
                        torch.from_numpy(xs).cuda(), torch.from_numpy(xt).cuda(), theta, ws)
    assert np.abs(got.cpu().numpy() - ref).max() <= 1e-5 * max(1, np.abs(ref).max())


STEP_CASES = [
    # style, dist, D, L, K, act, B, nv, loss kwargs, directed
    ('dist', 'pcd', 4096, 20, 3, None, 512, 58.388599, dict(), False),
    ('dist', 'pcd', 4096, 10, 4, None, 100, 58.388599, dict(reg_const=1e-3), False),
    ('dist', 'pcd', 2048, 20, 5, None, 96, 1.0, dict(), False),
    ('dist', 'pcd', 256, 5, 1, None, 33, 1.0, dict(reg_const=1e-2), False),
    ('cfl', 'pcd', 1024, 64, 3, None, 128, 31.9098,
     dict(pos_weight=0.0625, lambda_m=0.5, reg_const=5e-4), False),
    ('cfl', 'pcd', 1024, 16, 2, None, 64, 31.9098, dict(lambda_m=0.5), True),
    ('cfl', 'monomer', 1024, 64, 3, None, 128, 31.9098, dict(pos_weight=0.0625), False),
    ('cfl', 'monomer', 512, 12, 2, 'tanh', 40, 1.0, dict(reg_const=1e-3), True),
    ('cfl', 'siamese', 1024, 256, 1, None, 64, 31.9098,
     dict(use_threshold=False, caffe_margin=100.0, pos_weight=0.0625), False),
    ('cfl', 'siamese', 512, 32, 1, 'sigmoid', 48, 1.0, dict(), False),
    ('cfl', 'pcd', 512, 20, 4, 'relu', 64, 1.0, dict(use_threshold=False, lambda_m=0.3), False),
    # BASELINE.json full-size configs 4 and 3
    ('cfl', 'pcd', 2048, 20, 5, None, 1024, 1.0, dict(pos_weight=0.25), False),
    ('cfl', 'siamese', 1024, 256, 1, None, 512, 31.9098,
     dict(use_threshold=False, caffe_margin=100.0, pos_weight=0.0625), False),
    # large batches (narrow inputs: exact-fp32 forward; the bf16x3 forward needs >= 256 work units), 16384 rows also the register row math
    ('dist', 'pcd', 512, 20, 3, None, 4096, 58.388599, dict(), False),
    ('dist', 'pcd', 256, 10, 4, None, 8192, 58.388599, dict(reg_const=1e-3), False),
    ('cfl', 'pcd', 384, 20, 3, None, 4096, 31.9098, dict(pos_weight=0.25), False),
    # training batches from 1536 rows on: bf16x3 forward with x kept in cache for the weight gradient (cfl_proj_x3_keep_kernel),
    # half-tile weight gradient with the rows split in two
    ('dist', 'pcd', 4096, 20, 3, None, 1600, 58.388599, dict(), False),
    ('cfl', 'pcd', 2048, 20, 3, None, 2048, 31.9098, dict(pos_weight=0.25, reg_const=1e-3), False),
    # edge shapes: single row, odd batch, minimum D, many prototypes (40 column tiles)
    ('dist', 'pcd', 64, 3, 2, None, 1, 1.0, dict(), False),
    ('cfl', 'monomer', 64, 5, 3, None, 3, 1.0, dict(reg_const=1e-3), False),
    ('cfl', 'pcd', 256, 40, 16, None, 70, 1.0, dict(), False),
]


# Gradient bars.  TWO assertions per STEP_CASE and tensor:
#  * HARD_GRAD_CEILING -- a FIXED fp32-level ceiling (5e-6 of the tensor's scale), independent of anything the path under test
#    has ever produced: this is the parity bar.  Every case is held to it whatever the recorded file says.
#  * a regression guard: twice the case's worst error as recorded ONCE in tests/golden/step_grad_errors.json (commit 8ef9e1b,
#    round 4; 2.5e-7 ... 1.5e-6, floor GRAD_FLOOR).  The kernels are bit-reproducible run to run, so the factor 2 only covers
#    compiler / plan changes; re-recording the file can loosen THIS bar, never the ceiling.
GRAD_ERR_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'step_grad_errors.json')
GRAD_FLOOR = 1e-6
HARD_GRAD_CEILING = 5e-6
UNRECORDED_GRAD_CAP = HARD_GRAD_CEILING      # a STEP_CASE without a recorded entry: the ceiling alone


def _case_id(style, dist, D, L, K, act, B, lkw, directed):
    kw = ','.join('%s=%g' % (k, float(v)) for k, v in sorted(lkw.items()))
    return '%s-%s-D%d-L%d-K%d-%s-B%d-%s%s' % (style, dist, D, L, K, act or 'linear', B, kw, '-directed' if directed else '')


def _grad_caps():
    try:
        with open(GRAD_ERR_FILE) as f:
            return json.load(f)['cases']
    except (OSError, ValueError, KeyError):
        return {}


@pytest.mark.parametrize('style,dist,D,L,K,act,B,nv,lkw,directed', STEP_CASES)
def test_step_fwd_bwd(style, dist, D, L, K, act, B, nv, lkw, directed):
    rng = np.random.RandomState(4321)
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style=style, act_type=act)
    lcfg = O.LossCfg(**lkw)
    p = _mk(cfg, rng)
    pd = _mk(cfg, rng) if directed else None
    thr = 0.9 if dist != 'siamese' else 40.0
    batch = tuple(_inputs(rng, B, D, nv / 4) for _ in range(4))
    b64 = tuple(b.astype(np.float64) / nv for b in batch)
    sc, g, gd, dthr, dthr_aux = O.train_step_loss_and_grads(
        cfg, lcfg, _to64(p), np.float64(thr), b64, _to64(pd))

    sh = _shape(cfg, directed)
    theta = H.pack_theta(sh, p, pd, thr, 'cuda')
    grad = torch.full_like(theta, float('nan'))
    scal = torch.zeros(H.S_COUNT, device='cuda')
    ws = torch.empty(H.workspace_bytes(sh, B, 2) // 4, dtype=torch.float32, device='cuda')
    ws.fill_(float('nan'))            # poison: nothing may depend on stale scratch
    H.pair_step_fwd_bwd(sh, H.make_norm(1.0 / nv), H.make_loss(**lkw),
                        [torch.from_numpy(b).cuda() for b in batch], theta, grad, scal, ws)
    torch.cuda.synchronize()
    s = dict(zip(H.SCALAR_NAMES, scal.cpu().numpy().astype(np.float64)))
    for k in ('total', 'reg', 'thres', 'loss_pos', 'loss_neg', 'cd', 'accuracy',
              'mean_d_pos', 'mean_d_neg'):
        ref = float(sc[k])
        assert abs(s[k] - ref) <= 1e-5 * max(1.0, abs(ref)), (k, s[k], ref)
    assert not torch.isnan(grad).any()
    gp, gpd, gthr = H.unpack_theta(sh, grad)
    want_thr = dthr if lcfg.use_threshold else dthr_aux
    assert abs(gthr - want_thr) <= 1e-5 * max(1.0, abs(want_thr)), (gthr, want_thr)
    # per-tensor relative bound; a tensor whose whole gradient is a cancellation residue (>1000x below the
    # largest gradient of the step) is held to that absolute floor instead
    gmax = max([float(np.abs(np.asarray(v)).max()) for ref in (g, gd) if ref is not None for v in ref.values()] + [0.0])
    cid = _case_id(style, dist, D, L, K, act, B, lkw, directed)
    record = os.environ.get('CFL_RECORD_GRAD_ERRORS')
    observed = {}
    for enc, got, ref in (('', gp, g), ('dst:', gpd, gd)):
        if ref is None:
            continue
        for k in got:
            r = np.asarray(ref.get(k, np.zeros_like(got[k])), dtype=np.float64)
            scale = max(np.abs(r).max(), 1e-3 * gmax, 1e-6)
            observed[enc + k] = float(np.abs(got[k] - r).max() / scale)
    worst = max(observed.values())
    if record:
        try:
            with open(record) as f:
                doc = json.load(f)
        except (OSError, ValueError):
            doc = {'what': 'worst per-tensor |grad_hip - grad_fp64| / tensor scale of tests/test_hip_parity.py::test_step_fwd_bwd',
                   'cases': {}}
        doc['cases'][cid] = {'worst': worst, 'per_tensor': observed}
        with open(record, 'w') as f:
            json.dump(doc, f, indent=1, sort_keys=True)
        return
    caps = _grad_caps()
    fuzz_cap = os.environ.get('CFL_FUZZ_GRAD_CAP')     # tools/fuzz_parity.py: random shapes have no record; one generous fp32-level cap
    if cid not in caps and fuzz_cap:
        caps = {cid: {'worst': 0.5 * float(fuzz_cap)}}
    # a case without a record (a newly added shape) is held to a fixed fp32-level cap instead of failing: record it with
    # CFL_RECORD_GRAD_ERRORS=<file> to get its own bar (2 x observed)
    cap = max(2.0 * caps[cid]['worst'], GRAD_FLOOR) if cid in caps else UNRECORDED_GRAD_CAP
    for k, e in observed.items():
        if not fuzz_cap:
            assert e <= HARD_GRAD_CEILING, (cid, k, e, 'fixed fp32-level ceiling')     # the parity bar: independent of the path under test
        assert e <= cap, (cid, k, e, cap)                                               # regression guard (recorded once)


def test_full_size_properties():
    """Size-independent properties at the headline size (B=512, D=4096, K=3, L=20):
    (i) the loss is invariant under a permutation of the rows inside the positive and the
    negative group and the gradient changes only by summation order; (ii) the gradient of the
    batch equals the mean of the gradients of its two half-batches (the identity the
    data-parallel all-reduce relies on); (iii) scoring is row-wise: scores of a batch equal
    the scores of its rows taken one chunk at a time."""
    rng = np.random.RandomState(2026)
    B, D, K, L, nv = 512, 4096, 3, 20, 58.388599
    cfg = O.EncoderCfg(D=D, L=L, K=K)
    p = _mk(cfg, rng)
    sh = _shape(cfg)
    theta = H.pack_theta(sh, p, None, 0.8, 'cuda')
    norm, loss = H.make_norm(1.0 / nv), H.make_loss()
    batch = [torch.from_numpy(_inputs(rng, B, D, nv / 4)).cuda() for _ in range(4)]

    def run(b):
        n = b[0].shape[0]
        g = torch.empty_like(theta)
        sc = torch.zeros(H.S_COUNT, device='cuda')
        ws = torch.empty(H.workspace_bytes(sh, n, 2) // 4, dtype=torch.float32, device='cuda')
        H.pair_step_fwd_bwd(sh, norm, loss, b, theta, g, sc, ws)
        return g, sc.cpu().numpy()
    g0, s0 = run(batch)
    pp, pn = torch.randperm(B, device='cuda'), torch.randperm(B, device='cuda')
    g1, s1 = run([batch[0][pp].contiguous(), batch[1][pp].contiguous(),
                  batch[2][pn].contiguous(), batch[3][pn].contiguous()])
    assert abs(s0[0] - s1[0]) <= 2e-6 * max(1.0, abs(s0[0]))
    gmax = float(g0.abs().max())
    assert float((g0 - g1).abs().max()) <= 1e-5 * gmax
    h = B // 2
    ga, _ = run([x[:h].contiguous() for x in batch])
    gb, _ = run([x[h:].contiguous() for x in batch])
    assert float((0.5 * (ga + gb) - g0).abs().max()) <= 1e-5 * gmax
    ws = torch.empty(H.workspace_bytes(sh, B, 1) // 4, dtype=torch.float32, device='cuda')
    full = H.pair_scores(sh, norm, batch[0], batch[1], theta, ws)
    parts = torch.cat([H.pair_scores(sh, norm, batch[0][i:i + 100].contiguous(),
                                     batch[1][i:i + 100].contiguous(), theta, ws)
                       for i in range(0, B, 100)])
    assert torch.equal(full, parts)          # bit-exact: no cross-row arithmetic


def test_adam_tf_flat():
    rng = np.random.RandomState(0)
    n = 64 * 1000
    th = rng.randn(n).astype(np.float32)
    m = (0.1 * rng.randn(n)).astype(np.float32)
    v = np.abs(0.01 * rng.randn(n)).astype(np.float32)
    g = rng.randn(n).astype(np.float32)
    ref = O.adam_tf_flat(th.astype(np.float64), m.astype(np.float64), v.astype(np.float64),
                         0.5 * g.astype(np.float64), 3e-3, 0.9, 0.999, 1e-8)
    t, tm, tv, tg = (torch.from_numpy(a.copy()).cuda() for a in (th, m, v, g))
    H.adam_tf(t, tm, tv, tg, 3e-3, 0.9, 0.999, 1e-8, grad_scale=0.5)
    for got, r in zip((t, tm, tv), ref):
        assert np.abs(got.cpu().numpy() - r).max() <= 1e-6 * max(1, np.abs(r).max())


def test_gather_rows_bit_exact():
    rng = np.random.RandomState(3)
    table = rng.randn(1000, 4096).astype(np.float32)
    idx = rng.randint(0, 1000, size=777).astype(np.int64)
    got = H.gather_rows(torch.from_numpy(table).cuda(), torch.from_numpy(idx).cuda())
    assert np.array_equal(got.cpu().numpy(), table[idx])


@pytest.mark.parametrize('style,dist,K,L,lkw', [
    ('dist', 'pcd', 3, 20, dict()),
    ('cfl', 'pcd', 3, 16, dict(pos_weight=0.25, lambda_m=0.5, reg_const=5e-4)),
    ('cfl', 'monomer', 2, 12, dict()),
])
def test_training_trajectory(style, dist, K, L, lkw):
    """Per-step loss within 1e-5 of the fp64 oracle over the first 100 Adam steps on identical
    batches, and final scores / AUC within 1e-4 (north star; SURVEY 8(d) "AUC check")."""
    rng = np.random.RandomState(99)
    D, B, nv, steps = 1024, 128, 8.0, 100
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style=style)
    lcfg = O.LossCfg(**lkw)
    p = O.init_encoder_params(cfg, rng, np.float32)
    tr = O.OracleTrainer(cfg, lcfg, lr=1e-3, dtype=np.float64, params=_to64(p))
    A = (rng.randn(D, D) * 0.03).astype(np.float32)
    def batch():
        xs = _inputs(rng, B, D, 2.0)
        xp = np.abs(xs @ A + 0.1 * rng.randn(B, D)).astype(np.float32)
        return xs, xp, _inputs(rng, B, D, 2.0), _inputs(rng, B, D, 2.0)
    sh = _shape(cfg)
    theta = H.pack_theta(sh, p, None, 1e-6, 'cuda')
    m = torch.zeros_like(theta)
    v = torch.zeros_like(theta)
    grad = torch.zeros_like(theta)
    scal = torch.zeros(H.S_COUNT, device='cuda')
    ws = torch.empty(H.workspace_bytes(sh, B, 2) // 4, dtype=torch.float32, device='cuda')
    norm, loss = H.make_norm(1.0 / nv), H.make_loss(**lkw)
    b1p, b2p = np.float32(0.9), np.float32(0.999)
    for it in range(steps):
        b = batch()
        sc = tr.step(tuple(x.astype(np.float64) / nv for x in b))
        H.pair_step_fwd_bwd(sh, norm, loss, [torch.from_numpy(x).cuda() for x in b],
                            theta, grad, scal, ws)
        lr_t = np.float32(1e-3) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p)
        H.adam_tf(theta, m, v, grad, lr_t, 0.9, 0.999, 1e-8)
        b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
        got = float(scal[0].item())
        assert abs(got - sc['total']) <= 1e-5 * max(1.0, abs(sc['total'])), (it, got, sc['total'])
    xs, xp, xn1, xn2 = batch()
    ws2 = torch.empty(H.workspace_bytes(sh, B, 1) // 4, dtype=torch.float32, device='cuda')
    sp = H.pair_scores(sh, norm, torch.from_numpy(xs).cuda(), torch.from_numpy(xp).cuda(), theta, ws2).cpu().numpy()
    sn = H.pair_scores(sh, norm, torch.from_numpy(xn1).cuda(), torch.from_numpy(xn2).cuda(), theta, ws2).cpu().numpy()
    rp = tr.scores(xs.astype(np.float64) / nv, xp.astype(np.float64) / nv)
    rn = tr.scores(xn1.astype(np.float64) / nv, xn2.astype(np.float64) / nv)
    assert np.abs(sp - rp).max() <= 1e-4 * max(1, np.abs(rp).max())
    ev_h, ev_o = O.dist_eval(sp, sn), O.dist_eval(rp, rn)
    assert abs(ev_h['auc'] - ev_o['auc']) <= 1e-4


def test_errors_are_reported_not_thrown():
    sh = H.make_shape(4096, 20, 3)
    ws = torch.empty(16, dtype=torch.float32, device='cuda')
    theta = torch.zeros(H.layout(sh).total, device='cuda')
    x = torch.zeros(8, 4096, device='cuda')
    with pytest.raises(H.CflHipError, match='workspace'):
        H.pair_scores(sh, H.make_norm(), x, x, theta, ws)
    with pytest.raises(H.CflHipError, match='no CPU fallback'):
        H.pair_scores(sh, H.make_norm(), x.cpu(), x.cpu(), theta, ws)


@pytest.mark.parametrize('style,dist,K,L', [('dist', 'pcd', 3, 20), ('cfl', 'monomer', 2, 12),
                                            ('cfl', 'siamese', 1, 32)])
def test_fused_train_step_equals_fwd_bwd_plus_adam(style, dist, K, L):
    """cfl_pair_train_step (Adam fused into the last kernel) == cfl_pair_step_fwd_bwd
    followed by cfl_adam_tf, including the weight-norm gain snapshot."""
    rng = np.random.RandomState(77)
    D, B, nv = 512, 96, 4.0
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style=style)
    p = _mk(cfg, rng)
    sh = _shape(cfg)
    lkw = dict(reg_const=1e-3, pos_weight=0.5)
    norm, loss = H.make_norm(1.0 / nv), H.make_loss(**lkw)
    batch = [torch.from_numpy(_inputs(rng, B, D, 1.0)).cuda() for _ in range(4)]
    ws = torch.empty(H.workspace_bytes(sh, B, 2) // 4, dtype=torch.float32, device='cuda')
    res = []
    for fused in (False, True):
        theta = H.pack_theta(sh, p, None, 0.5, 'cuda')
        m = torch.full_like(theta, 0.01)
        v = torch.full_like(theta, 0.02)
        grad = torch.zeros_like(theta)
        scal = torch.zeros(H.S_COUNT, device='cuda')
        for _ in range(3):
            if fused:
                H.pair_train_step(sh, norm, loss, batch, theta, m, v, grad, scal, ws, 2e-3, 0.9, 0.999)
            else:
                H.pair_step_fwd_bwd(sh, norm, loss, batch, theta, grad, scal, ws)
                H.adam_tf(theta, m, v, grad, 2e-3, 0.9, 0.999)
        res.append([t.cpu().numpy() for t in (theta, m, v, grad, scal)])
    for a, b in zip(*res):
        assert np.allclose(a, b, rtol=1e-6, atol=1e-9)


def test_bf16x3_matrix_core_path_is_fp32_equivalent(monkeypatch):
    """The weight-gradient contraction runs on the bf16 matrix cores with every fp32 operand split exactly
    in three bf16 values (6 partial products).  Against the fp64 oracle its error must be the same size as
    that of the k-ordered fp32 MFMA kernel (CFL_EXACT_FP32=1); the forward is the fp32 kernel either way."""
    from cfl import hipabi as H
    from cfl.engine import PairEngine
    B, D, K, L = 256, 1024, 3, 20
    cfg = O.EncoderCfg(D=D, L=L, K=K)
    rng = np.random.RandomState(7)
    params = O.init_encoder_params(cfg, rng, np.float32)
    batch = [np.abs(rng.randn(B, D)).astype(np.float32) * 13 for _ in range(4)]
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    xs = tuple(b.astype(np.float64) / 58.388599 for b in batch)
    sc, grads, _, _, _ = O.train_step_loss_and_grads(cfg, O.LossCfg(), p64, 1e-6, xs)
    dev = [torch.from_numpy(b).cuda() for b in batch]
    errs = {}
    for mode, env in (('x3', '0'), ('fp32', '1')):
        monkeypatch.setenv('CFL_EXACT_FP32', env)
        H.reload_env()            # launch plans (incl. this switch) are cached per process
        eng = PairEngine(D, L, K, norm=H.make_norm(1 / 58.388599), params=params, batch_size=B)
        eng.fwd_bwd(dev)
        g, _, _ = H.unpack_theta(eng.shape, eng.grad)
        e = {'loss': abs(eng.read_scalars()['total'] - sc['total'])}
        for k in ('outputs/W', 'proto/W'):
            e[k] = np.abs(g[k].astype(np.float64) - grads[k]).max() / np.abs(grads[k]).max()
        errs[mode] = e
    for k in ('outputs/W', 'proto/W'):
        assert errs['x3'][k] < 2e-6 and errs['fp32'][k] < 2e-6, errs
        assert errs['x3'][k] <= 2.0 * errs['fp32'][k] + 1e-7, errs
    assert errs['x3']['loss'] <= 1e-6 and errs['fp32']['loss'] <= 1e-6, errs
    monkeypatch.undo()
    H.reload_env()


def test_long_row_ranges_take_the_unstaged_gradient_kernel(monkeypatch):
    """The weight-gradient kernel stages the addresses of a workgroup's row range in LDS when the range has at most
    8192 rows, and computes them per 64-row group otherwise (cfl_grad_x3_longrange_kernel).  One row range over
    2 x 4700 rows (CFL_DEBUG_P=1) takes the second kernel: gradient against the fp64 oracle, and dense == indexed
    bit for bit in both kernels."""
    from cfl import hipabi as H
    from cfl.engine import PairEngine
    B, D, K, L = 4700, 128, 2, 6
    cfg = O.EncoderCfg(D=D, L=L, K=K)
    rng = np.random.RandomState(11)
    params = O.init_encoder_params(cfg, rng, np.float32)
    table = np.abs(rng.randn(3000, D)).astype(np.float32)
    idx = [rng.randint(0, 3000, size=B).astype(np.int32) for _ in range(4)]
    batch = [table[i] for i in idx]
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    sc, grads, _, _, _ = O.train_step_loss_and_grads(cfg, O.LossCfg(), p64, 1e-6,
                                                     tuple(b.astype(np.float64) / 4.0 for b in batch))
    tdev = torch.from_numpy(table).cuda()
    streams = H.IndexStreams.from_tensors([torch.from_numpy(i).cuda() for i in idx])
    dev = [torch.from_numpy(b).cuda() for b in batch]
    got = {}
    for mode, env in (('staged', '0'), ('long', '1')):
        monkeypatch.setenv('CFL_DEBUG_P', env)
        H.reload_env()
        e1 = PairEngine(D, L, K, norm=H.make_norm(0.25), params=params, batch_size=B)
        e2 = PairEngine(D, L, K, norm=H.make_norm(0.25), params=params, batch_size=B)
        e1.fwd_bwd(dev)
        e2.fwd_bwd((tdev, streams))
        assert torch.equal(e1.grad, e2.grad) and torch.equal(e1.scalars, e2.scalars), mode
        g, _, _ = H.unpack_theta(e1.shape, e1.grad)
        for k in ('outputs/W', 'proto/W'):
            err = np.abs(g[k].astype(np.float64) - grads[k]).max() / np.abs(grads[k]).max()
            # P = 1 accumulates all 9400 rows in one fp32 chain per wave: a few 1e-6 of rounding
            assert err < 1e-5, (mode, k, err)
        got[mode] = e1.grad.clone()
    assert float((got['staged'] - got['long']).abs().max()) <= 1e-5 * float(got['long'].abs().max())
    monkeypatch.undo()
    H.reload_env()


@pytest.mark.parametrize('B,D,K,L,reg,P', [(512, 4096, 3, 20, 0.0, 0), (128, 1024, 5, 12, 1e-3, 0),
                                            # weight-normalised heads (P < 0 marks them; |P| - 1 = forced split): the tail
                                            # also waits for the c_j column sums of the launch's reduction blocks
                                            (1024, 2048, 5, 20, 1e-3, -1), (128, 1024, 3, 12, 1e-3, -1), (512, 576, 2, 7, 0.0, -1),
                                            (512, 1024, 3, 20, 0.0, -5), (16, 256, 1, 7, 1e-3, -1),
                                            (16, 256, 1, 7, 1e-3, 0), (1024, 2048, 3, 20, 0.0, 0),
                                            # forced row splits: short row ranges, many publishers per tile (this is
                                            # where separate inline-asm store / wait statements lost half a float4)
                                            (512, 4096, 3, 20, 0.0, 4), (512, 4096, 3, 20, 0.0, 8),
                                            (1024, 2048, 3, 20, 0.0, 8), (2048, 1024, 3, 20, 1e-3, 0),
                                            # d-tile counts that are not multiples of 8: the row ranges of a tile run on
                                            # DIFFERENT XCDs (separate L2s), the hand-off crosses them
                                            (512, 576, 3, 20, 0.0, 0), (1024, 320, 2, 12, 1e-3, 4), (512, 4160, 3, 20, 0.0, 0)])
def test_fused_gradient_tail_equals_finalize_kernel(B, D, K, L, reg, P, monkeypatch):
    """The `Dist` step finishes gradient + Adam inside the weight-gradient launch (pairs of row-range workgroups
    hand their partial tile over with sc1 stores / loads, the row-reduction blocks finish biases, threshold and
    scalars): 3 launches.  CFL_DEBUG_NOFUSE=1 keeps the 4-launch form with the finalize kernel.  Same arithmetic in
    the same order => bit-identical weights, Adam slots, gradient and scalars, step after step (300 steps: the
    hand-off is exercised ~40 000 times per configuration, under the uneven load of a running training loop)."""
    from cfl import hipabi as H
    from cfl.engine import PairEngine
    rng = np.random.RandomState(5)
    wn = P < 0
    if wn:
        P = -P - 1
    cfg = O.EncoderCfg(D=D, L=L, K=K, style='cfl' if wn else 'dist')
    params = O.init_encoder_params(cfg, rng, np.float32)
    if wn:   # gains away from 1 so that the correction term is exercised
        params = {k: (v * (1.0 + 0.3 * rng.rand(*v.shape)).astype(np.float32) if k.endswith('/g') else v)
                  for k, v in params.items()}
    pool = [[torch.from_numpy(np.abs(rng.randn(B, D)).astype(np.float32) * 3).cuda() for _ in range(4)]
            for _ in range(3)]
    res = {}
    if P:
        monkeypatch.setenv('CFL_DEBUG_P', str(P))
    # the same forward arithmetic in both modes: the bf16x3 projection the fused step uses with its kept planes, with a
    # per-call split of the weights in the four-launch form (which updates theta in the finalize launch and keeps none)
    monkeypatch.setenv('CFL_DEBUG_PROJ_BX3', '1')
    for mode in ('fused', 'finalize'):
        monkeypatch.setenv('CFL_DEBUG_NOFUSE', '0' if mode == 'fused' else '1')
        H.reload_env()
        eng = PairEngine(D, L, K, weight_norm=cfg.weight_norm, has_bias=cfg.has_bias, norm=H.make_norm(1 / 8.0),
                         loss=H.make_loss(reg_const=reg), params=params, batch_size=B)
        H.profile_enable(True)
        eng.step(pool[0])
        torch.cuda.synchronize()
        H.profile_enable(False)
        kinds = set(H.profile_read())
        assert ('finalize' in kinds) == (mode == 'finalize'), kinds
        snaps = []
        for it in range(300):
            eng.step(pool[it % 3])
            if it in (0, 1, 7, 299):
                snaps.append([t.clone() for t in (eng.theta, eng.m, eng.v, eng.grad, eng.scalars)])
        # the forward/backward-only entry point (data-parallel path): gradient without Adam
        eng.fwd_bwd(pool[1])
        snaps.append([eng.grad.clone(), eng.scalars.clone(), eng.theta.clone()])
        res[mode] = snaps
    monkeypatch.undo()
    H.reload_env()
    for si, (a, b) in enumerate(zip(res['fused'], res['finalize'])):
        for ti, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), (si, ti, (x != y).nonzero().flatten()[:8].tolist(), x.flatten()[(x != y).flatten()][:4].tolist(),
                                       y.flatten()[(x != y).flatten()][:4].tolist())
    assert np.isfinite(res['fused'][-1][1].cpu().numpy()).all()


def test_step_windows_equal_single_indexed_steps(tmp_path):
    """PairEngine.step_windows (one library call for the iterations between two scalar read-backs) == the same
    iterations as single indexed steps: identical index stream (heads, data_switch coin flips, epoch wraps handled
    by the caller) and bit-identical training, including TF-Adam's float32 power accumulators."""
    from cfl import hipabi as H
    from cfl import input_data
    from cfl.engine import PairEngine
    from cfl.synthetic import make_dataset
    root = make_dataset(str(tmp_path / 'toy'), D=256, n_items=300, n_pos=900, n_neg=700, k=2, latent=6, seed=1)
    path = str(tmp_path / 'toy' / 'train')
    a = input_data.SemiDataSet(path, input_size=256, data_switch=True, seed=9)
    b = input_data.SemiDataSet(path, input_size=256, data_switch=True, seed=9)
    ra, rb = input_data.ResidentFeatures(a), input_data.ResidentFeatures(b)
    rng = np.random.RandomState(0)
    cfg = O.EncoderCfg(D=256, L=6, K=3)
    params = O.init_encoder_params(cfg, rng, np.float32)
    mk = lambda: PairEngine(256, 6, 3, norm=H.make_norm(1 / 16.0), loss=H.make_loss(), params=params, lr=2e-3)
    ea, eb = mk(), mk()
    B, total, done = 64, 60, 0                       # 700 // 64 = 10 batches per epoch of the shorter list: wraps
    while done < total:
        win = ra.next_windows(B, min(7, total - done))
        if win is None:
            ea.step(ra.next_indexed(B))
            n = 1
        else:
            ea.step_windows(win)
            n = win.nsteps
        for _ in range(n):
            eb.step(rb.next_indexed(B))
        done += n
        assert torch.equal(ea.theta, eb.theta) and torch.equal(ea.scalars, eb.scalars)
        assert ea.beta1_power == eb.beta1_power and ea.beta2_power == eb.beta2_power
        assert ea.global_step == eb.global_step
    assert a.head_labeled_pos == b.head_labeled_pos and a.head_labeled_neg == b.head_labeled_neg
    assert np.array_equal(a.pairs_pos, b.pairs_pos) and np.array_equal(a.pairs_neg, b.pairs_neg)
    assert a._rng.rand() == b._rng.rand()
    # a window that runs past the end of a pair list is rejected before anything is launched (the index arrays
    # themselves would be read out of bounds; table indices are clamped, list positions cannot be)
    win = ra.next_windows(B, 2)
    while win is None:                 # (an epoch wrap is due: take it with a single step)
        ea.step(ra.next_indexed(B))
        win = ra.next_windows(B, 2)
    before = ea.theta.clone()
    with pytest.raises(H.CflHipError, match='past the pair lists'):
        H.pair_train_steps_idx(ea.shape, ea.norm, ea.loss, win.table, win.pos_pairs, win.neg_pairs,
                               win.pos_pairs.shape[0] - B, win.neg_head, win.batch_rows, win.shard_lo, win.rows, None, 2,
                               ea.theta, ea.m, ea.v, ea.grad, ea.scalars, ea._workspace(win.rows, 2),
                               np.float32(ea.lr), ea.beta1, ea.beta2, ea.eps, ea.beta1_power, ea.beta2_power)
    torch.cuda.synchronize()
    assert torch.equal(ea.theta, before)


@pytest.mark.parametrize('D,L,K,n,act_norm', [(4096, 20, 3, 8192, False), (2048, 20, 5, 4500, False), (1088, 12, 2, 3000, True),
                                              (4096, 64, 1, 2100, False)])
def test_projection_forms_agree(D, L, K, n, act_norm, monkeypatch):
    """The forms of the forward projection on large scoring calls (a wave owns several 128-d chunks):
    chunk-at-a-time (proj_body), streaming (proj_stream_body)
    and the bf16x3 form with LDS-shared W planes (cfl_proj_x3_kernel: the default from 4096 rows per side in scoring calls).  The
    streaming form must reproduce the chunk-at-a-time scores BIT FOR BIT (same k-ordered FMA chains, same summation
    order); the bf16x3 form accumulates eight exact partial products per
    32-d block, so it is held to fp32 rounding of the scores -- and additionally to an error
    against the float64 oracle no larger than twice the exact-fp32 form's -- and every form to 1e-5 of the oracle.
    Shapes: the dist_eval call (two jobs, 4 / 2 column tiles), 7 column tiles on the source side (jobs of 4 + 3
    tiles), D % 128 == 64 with an element-wise normaliser (the bf16x3 form declines it), a K = 1 model."""
    rng = np.random.RandomState(99)
    cfg = O.EncoderCfg(D=D, L=L, K=K)
    p = _mk(cfg, rng)
    sh = _shape(cfg)
    thr, nv = 0.41, 58.388599
    theta = H.pack_theta(sh, p, None, thr, 'cuda')
    xs = torch.from_numpy(_inputs(rng, n, D, nv / 4)).cuda()
    xt = torch.from_numpy(_inputs(rng, n, D, nv / 4)).cuda()
    norm = H.make_norm(1.0 / nv, -0.05, 0.0, 0.9) if act_norm else H.make_norm(1.0 / nv)

    def run(stream, x3):
        monkeypatch.setenv('CFL_DEBUG_PROJ_STREAM', str(stream))
        monkeypatch.setenv('CFL_DEBUG_PROJ_X3', str(x3))
        H.reload_env()
        ws = torch.full((H.workspace_bytes(sh, n, 1) // 4,), float('nan'), dtype=torch.float32, device='cuda')
        out = H.pair_scores(sh, norm, xs, xt, theta, ws).clone()
        torch.cuda.synchronize()
        return out
    try:
        classic, stream, x3 = run(-1, -1), run(1, -1), run(-1, 1)
    finally:
        for k in ('CFL_DEBUG_PROJ_STREAM', 'CFL_DEBUG_PROJ_X3'):
            monkeypatch.delenv(k)
        H.reload_env()
    assert torch.equal(classic, stream)
    scale = max(1.0, float(classic.abs().max()))
    assert float((classic - x3).abs().max()) <= 4e-6 * scale
    f = (lambda x: O.normalize_v2(x.cpu().numpy().astype(np.float64), scale=1.0 / nv, mean=0.05, norm=1.0, clip_min=0.0,
                                  clip_max=0.9)) if act_norm else (lambda x: x.cpu().numpy().astype(np.float64) / nv)
    sub = slice(0, 1500)
    ref = O.pair_scores(cfg, _to64(p), np.float64(thr), f(xs[sub]), f(xt[sub]))
    err = {}
    for name, got in (('classic', classic), ('x3', x3)):
        err[name] = np.abs(got[sub].cpu().numpy() - ref).max()
        assert err[name] <= 1e-5 * max(1.0, np.abs(ref).max()), (name, err[name])
    assert err['x3'] <= max(2.0 * err['classic'], 2e-6 * scale), err


@pytest.mark.parametrize('B,D,L,wn,lkw,P', [
    (512, 1024, 256, True, dict(use_threshold=False, caffe_margin=100.0, pos_weight=0.0625), 0),     # BASELINE config 3
    (128, 1024, 40, True, dict(reg_const=1e-3), 0), (256, 576, 20, False, dict(lambda_m=0.5), 0),
    (512, 1024, 64, True, dict(use_threshold=False, caffe_margin=100.0), 4), (16, 256, 7, False, dict(reg_const=1e-3), 0)])
def test_fused_gradient_tail_siamese_equals_finalize_kernel(B, D, L, wn, lkw, P, monkeypatch):
    """Siamese models: both sides project through ONE head, so a weight tile receives 2 P row ranges (side 0's
    workgroups only publish, the last row range of side 1 finishes: 2P - 1 arrivals, side 0's slabs summed first) and
    the bias / gain / c_j column sums come from dual reduction ranges that cover both sides.  Bit-identical to the
    four-launch form with the finalize kernel over 300 steps, weight-normalised (config 3: L=256, hinge margin 100,
    separate threshold Adam) and plain."""
    from cfl.engine import PairEngine
    rng = np.random.RandomState(6)
    cfg = O.EncoderCfg(D=D, L=L, K=1, dist_type='siamese', style='cfl' if wn else 'dist')
    params = O.init_encoder_params(cfg, rng, np.float32)
    if wn:
        params = {k: (v * (1.0 + 0.3 * rng.rand(*v.shape)).astype(np.float32) if k.endswith('/g') else v)
                  for k, v in params.items()}
    pool = [[torch.from_numpy(np.abs(rng.randn(B, D)).astype(np.float32) * 3).cuda() for _ in range(4)]
            for _ in range(3)]
    res = {}
    if P:
        monkeypatch.setenv('CFL_DEBUG_P', str(P))
    # the same forward arithmetic in both modes: the bf16x3 projection the fused step uses with its kept planes, with a
    # per-call split of the weights in the four-launch form (which updates theta in the finalize launch and keeps none)
    monkeypatch.setenv('CFL_DEBUG_PROJ_BX3', '1')
    for mode in ('fused', 'finalize'):
        monkeypatch.setenv('CFL_DEBUG_NOFUSE', '0' if mode == 'fused' else '1')
        H.reload_env()
        eng = PairEngine(D, L, 1, 'siamese', weight_norm=cfg.weight_norm, has_bias=cfg.has_bias, norm=H.make_norm(1 / 8.0),
                         loss=H.make_loss(**lkw), params=params, thr=30.0 if lkw.get('caffe_margin') else 0.5, batch_size=B)
        H.profile_enable(True)
        eng.step(pool[0])
        torch.cuda.synchronize()
        H.profile_enable(False)
        kinds = set(H.profile_read())
        assert ('finalize' in kinds) == (mode == 'finalize'), kinds
        snaps = []
        for it in range(300):
            eng.step(pool[it % 3])
            if it in (0, 1, 7, 299):
                snaps.append([t.clone() for t in (eng.theta, eng.m, eng.v, eng.grad, eng.scalars)])
        eng.fwd_bwd(pool[1])
        snaps.append([eng.grad.clone(), eng.scalars.clone(), eng.theta.clone()])
        res[mode] = snaps
    monkeypatch.undo()
    H.reload_env()
    for si, (a, b) in enumerate(zip(res['fused'], res['finalize'])):
        for ti, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), (si, ti, (x != y).nonzero().flatten()[:8].tolist(), x.flatten()[(x != y).flatten()][:4].tolist(),
                                       y.flatten()[(x != y).flatten()][:4].tolist())
    assert np.isfinite(res['fused'][-1][1].cpu().numpy()).all()


def test_scores_of_a_labeled_batch_in_one_call():
    """cfl_pair_scores_idx4 (positive pairs then negative pairs of an indexed labeled batch, one projection + row-math
    launch pair: the validation fetch of the training loop) == two cfl_pair_scores_idx calls, to fp32 rounding (the d
    split of the projection follows the row count, so the slice sums may associate differently)."""
    from cfl.engine import PairEngine
    rng = np.random.RandomState(12)
    D, L, K, n, rows = 1024, 20, 3, 300, 5000
    cfg = O.EncoderCfg(D=D, L=L, K=K)
    eng = PairEngine(D, L, K, norm=H.make_norm(1 / 16.0), loss=H.make_loss(), params=_mk(cfg, rng))
    table = torch.from_numpy(np.abs(rng.randn(rows, D)).astype(np.float32) * 4).cuda()
    idx = [torch.from_numpy(rng.randint(0, rows, size=n).astype(np.int32)).cuda() for _ in range(4)]
    streams = H.IndexStreams.from_tensors(idx)
    both = eng.scores_pos_neg(table, streams).clone()
    sp = eng.scores(table, streams.pair(0)).clone()
    sn = eng.scores(table, streams.pair(1)).clone()
    ref = torch.cat([sp, sn])
    assert both.shape == (2 * n,)
    assert float((both - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    # and against the oracle, through the dense rows the indices select
    x = [table[i.long()].cpu().numpy().astype(np.float64) / 16.0 for i in idx]
    p, _, thr = H.unpack_theta(eng.shape, eng.theta)
    for lo, (a, b) in ((0, (x[0], x[1])), (n, (x[2], x[3]))):
        want = np.asarray(O.pair_scores(cfg, _to64(p), np.float64(thr), a, b)).reshape(-1)
        got = both[lo:lo + n].cpu().numpy()
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_training_forward_form_follows_the_batch_size(monkeypatch):
    """The fused training step keeps the bf16 planes of theta (CflThetaPlanes) and projects on the bf16 matrix cores:
    chunk-at-a-time form (cfl_proj_bx3_kernel) at the headline batch, LDS-shared form (cfl_proj_x3_kernel) from 2560 rows per call
    (round 5; 3072 until then).  Only the FIRST step of an engine splits the weights in a launch of its own (the library's profile lists it
    under `colnorm`); from the second step on the Adam tail has written the planes.  Against the exact-fp32 forward
    (CFL_DEBUG_PROJ_BX3=-1 / CFL_DEBUG_PROJ_X3=-1) the step's scalars move by fp32 rounding only."""
    from cfl.engine import PairEngine
    rng = np.random.RandomState(3)
    cfg = O.EncoderCfg(D=4096, L=20, K=3)
    params = O.init_encoder_params(cfg, rng, np.float32)
    out = {}
    for B, exact in ((512, False), (512, True), (1536, False), (1536, True)):
        if exact:
            monkeypatch.setenv('CFL_DEBUG_PROJ_X3', '-1')
            monkeypatch.setenv('CFL_DEBUG_PROJ_BX3', '-1')
        else:
            monkeypatch.delenv('CFL_DEBUG_PROJ_X3', raising=False)
            monkeypatch.delenv('CFL_DEBUG_PROJ_BX3', raising=False)
        H.reload_env()
        batch = [torch.from_numpy(np.abs(np.random.RandomState(B).randn(B, 4096)).astype(np.float32) * 13).cuda() for _ in range(4)]
        eng = PairEngine(4096, 20, 3, norm=H.make_norm(1 / 58.388599), loss=H.make_loss(), params=params, batch_size=B)
        kinds = []
        for _ in range(2):
            H.profile_enable(True)
            eng.step(batch)
            torch.cuda.synchronize()
            H.profile_enable(False)
            kinds.append(set(H.profile_read()))
        out[(B, exact)] = (kinds, eng.scalars.clone(), eng.grad.clone(), eng.planes.valid)
    monkeypatch.undo()
    H.reload_env()
    for B in (512, 1536):
        kinds, _, _, valid = out[(B, False)]
        assert 'colnorm' in kinds[0] and 'colnorm' not in kinds[1] and valid       # split once, then kept by the Adam tail
        kinds, _, _, valid = out[(B, True)]
        assert 'colnorm' not in kinds[0] and 'colnorm' not in kinds[1] and not valid
        a, b = out[(B, False)], out[(B, True)]
        assert float((a[1] - b[1]).abs().max()) <= 2e-6 * max(1.0, float(b[1].abs().max()))
        assert float((a[2] - b[2]).abs().max()) <= 2e-5 * float(b[2].abs().max())


@pytest.mark.parametrize('dist,B,D,K,L,wn,P', [
    ('pcd', 1536, 4096, 3, 20, False, 0),      # default plan: 32-d tiles, rows split in two (2048 < rows per side <= 6144)
    ('pcd', 512, 1024, 3, 20, True, 2), ('pcd', 256, 576, 2, 7, False, 4), ('pcd', 1024, 2048, 5, 20, True, 2),
    ('siamese', 512, 1024, 1, 256, True, 0),   # config 3's default: side 0's half tile published, side 1 finishes
    ('siamese', 256, 576, 1, 40, True, 2), ('siamese', 128, 1024, 1, 20, False, 4), ('siamese', 64, 256, 1, 7, False, 1)])
def test_half_tile_gradient_with_hand_off_equals_finalize_kernel(dist, B, D, K, L, wn, P, monkeypatch):
    """cfl_grad_x3_half_kernel with a row split and / or the siamese pairing: the first P - 1 row ranges (and all of
    side 0) publish their 32-d partial tile, the last row range finishes it (at most three published tiles per round
    trip, summed in the finalize kernel's order).  P > 0 forces half tiles with that split (CFL_DEBUG_GRAD_HALF=1 +
    CFL_DEBUG_P); P = 0 takes the plan's own choice, which must be the half-tile kernel for these shapes.
    Bit-identical to the four-launch form over 200 steps."""
    from cfl.engine import PairEngine
    rng = np.random.RandomState(8)
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style='cfl' if wn else 'dist')
    params = O.init_encoder_params(cfg, rng, np.float32)
    if wn:
        params = {k: (v * (1.0 + 0.3 * rng.rand(*v.shape)).astype(np.float32) if k.endswith('/g') else v)
                  for k, v in params.items()}
    pool = [[torch.from_numpy(np.abs(rng.randn(B, D)).astype(np.float32) * 3).cuda() for _ in range(4)]
            for _ in range(3)]
    res = {}
    if P:
        monkeypatch.setenv('CFL_DEBUG_P', str(P))
        monkeypatch.setenv('CFL_DEBUG_GRAD_HALF', '1')
    # the same forward arithmetic in both modes: the bf16x3 projection the fused step uses with its kept planes, with a
    # per-call split of the weights in the four-launch form (which updates theta in the finalize launch and keeps none)
    monkeypatch.setenv('CFL_DEBUG_PROJ_BX3', '1')
    for mode in ('fused', 'finalize'):
        monkeypatch.setenv('CFL_DEBUG_NOFUSE', '0' if mode == 'fused' else '1')
        H.reload_env()
        eng = PairEngine(D, L, K, dist, weight_norm=cfg.weight_norm, has_bias=cfg.has_bias, norm=H.make_norm(1 / 8.0),
                         loss=H.make_loss(reg_const=1e-3), params=params, batch_size=B)
        snaps = []
        for it in range(200):
            eng.step(pool[it % 3])
            if it in (0, 1, 7, 199):
                snaps.append([t.clone() for t in (eng.theta, eng.m, eng.v, eng.grad, eng.scalars)])
        eng.fwd_bwd(pool[1])
        snaps.append([eng.grad.clone(), eng.scalars.clone(), eng.theta.clone()])
        res[mode] = snaps
    monkeypatch.undo()
    H.reload_env()
    for si, (a, b) in enumerate(zip(res['fused'], res['finalize'])):
        for ti, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), (si, ti, (x != y).nonzero().flatten()[:8].tolist(), x.flatten()[(x != y).flatten()][:4].tolist(),
                                       y.flatten()[(x != y).flatten()][:4].tolist())
    assert np.isfinite(res['fused'][-1][1].cpu().numpy()).all()
    # and against the 64-d / row-split kernel (different summation order across the split: close, not identical)
    monkeypatch.setenv('CFL_DEBUG_GRAD_HALF', '-1')
    H.reload_env()
    eng = PairEngine(D, L, K, dist, weight_norm=cfg.weight_norm, has_bias=cfg.has_bias, norm=H.make_norm(1 / 8.0),
                     loss=H.make_loss(reg_const=1e-3), params=params, batch_size=B)
    eng.step(pool[0])
    monkeypatch.undo()
    H.reload_env()
    ref = res['fused'][0]
    assert float((eng.grad - ref[3]).abs().max()) <= 2e-5 * float(ref[3].abs().max())


@pytest.mark.parametrize('dist,directed,wn,B,D,L,K,act,lkw', [
    ('monomer', False, True, 128, 1024, 64, 3, None, dict(pos_weight=0.0625, reg_const=1e-3)),
    ('monomer', False, False, 64, 512, 12, 2, 'tanh', dict(reg_const=1e-3)),
    ('monomer', True, True, 96, 512, 20, 4, None, dict(lambda_m=0.5, reg_const=1e-3)),
    ('pcd', True, True, 128, 1024, 16, 2, None, dict(lambda_m=0.5, reg_const=1e-3)),
    ('pcd', True, False, 512, 576, 20, 3, None, dict()),
    ('siamese', True, True, 64, 512, 32, 1, 'sigmoid', dict(reg_const=1e-3)),
])
def test_fused_gradient_tail_monomer_and_directed_equal_finalize_kernel(dist, directed, wn, B, D, L, K, act, lkw, monkeypatch):
    """Every model runs the step in three launches: monomer (the gate head V[L][K] and its gains are finished by the
    reduction blocks that own their rows / column sums, the weight-norm correction sum recomputed in place) and directed
    encoders (the heads no side projects through get their L2-only gradient and Adam from element-wise blocks of the
    same launch).  Bit-identical to the four-launch form with the finalize kernel over 200 steps."""
    from cfl.engine import PairEngine
    rng = np.random.RandomState(8)
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style='cfl' if wn else 'dist', act_type=act)
    params = _mk(cfg, rng)
    params_dst = _mk(cfg, rng) if directed else None
    pool = [[torch.from_numpy(np.abs(rng.randn(B, D)).astype(np.float32) * 3).cuda() for _ in range(4)] for _ in range(3)]
    res = {}
    # the same forward arithmetic in both modes: the bf16x3 projection the fused step uses with its kept planes, with a
    # per-call split of the weights in the four-launch form (which updates theta in the finalize launch and keeps none)
    monkeypatch.setenv('CFL_DEBUG_PROJ_BX3', '1')
    for mode in ('fused', 'finalize'):
        monkeypatch.setenv('CFL_DEBUG_NOFUSE', '0' if mode == 'fused' else '1')
        H.reload_env()
        eng = PairEngine(D, L, K, dist, weight_norm=cfg.weight_norm, has_bias=cfg.has_bias, act_type=act, directed=directed,
                         norm=H.make_norm(1 / 8.0), loss=H.make_loss(**lkw), params=params, params_dst=params_dst,
                         thr=0.7, batch_size=B)
        H.profile_enable(True)
        eng.step(pool[0])
        torch.cuda.synchronize()
        H.profile_enable(False)
        kinds = set(H.profile_read())
        assert ('finalize' in kinds) == (mode == 'finalize'), kinds
        snaps = []
        for it in range(200):
            eng.step(pool[it % 3])
            if it in (0, 1, 7, 199):
                snaps.append([t.clone() for t in (eng.theta, eng.m, eng.v, eng.grad, eng.scalars)])
        eng.fwd_bwd(pool[1])
        snaps.append([eng.grad.clone(), eng.scalars.clone(), eng.theta.clone()])
        res[mode] = snaps
    monkeypatch.undo()
    H.reload_env()
    for si, (a, b) in enumerate(zip(res['fused'], res['finalize'])):
        for ti, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), (si, ti, (x != y).nonzero().flatten()[:8].tolist(), x.flatten()[(x != y).flatten()][:4].tolist(),
                                       y.flatten()[(x != y).flatten()][:4].tolist())
    assert np.isfinite(res['fused'][-1][1].cpu().numpy()).all()
