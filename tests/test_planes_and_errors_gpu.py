"""Round 4 (GPU): bf16 planes of theta kept by the fused training step, and loud in-launch failures.

* CflThetaPlanes: the tile finishers of the weight-gradient launch write the bf16 planes of the weights they update,
  so the bf16x3 projection of the next step needs no per-call split launch (cfl_wplanes_kernel).  The kept planes
  must equal a host restatement of the split of the updated theta BIT FOR BIT, and a training run with kept planes
  must equal the run that splits per call bit for bit -- every model family, both weight-gradient tile shapes.
* A training kernel that gives up on an in-launch hand-off sets the sticky error word scalars[CFL_S_ERROR]; the host
  raises CflHipError at its next read-back instead of training on NaN parameters (SURVEY 8(b) "Errors").
"""
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
    hipabi.lib()
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    H = hipabi
    yield


def _params(cfg, rng):
    p = O.init_encoder_params(cfg, rng, np.float32)
    for k in p:
        p[k] = (p[k] + 0.05 * rng.randn(*p[k].shape).astype(np.float32) * (0.1 if k.endswith('/W') else 1.0)).astype(np.float32)
    return p


def _host_planes(theta_host, head, D):
    """planes of one head restated on the host: level p of W[d][col] (exact truncation split, tests/test_bf16x3_split.py)
    at ushort 3 w + ((nt * Q + tq) * 3 + p) * 512 + lane * 8 + j, d = 32 tq + 8 (lane >> 4) + j, col = 16 nt + (lane & 15)"""
    wt = H.frag_to_wt(torch.from_numpy(theta_host[head.w:head.w + head.npad * D]), head.npad, D).numpy()   # [npad, D]
    def rne(x):      # float32 -> bf16 bits, round to nearest even (v_cvt_pk_bf16_f32)
        b = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
        return (((b + 0x7fff + ((b >> 16) & 1)) >> 16) & 0xffff).astype(np.uint32)

    def f32(hb):
        return (hb << 16).astype(np.uint32).view(np.float32)
    v = np.ascontiguousarray(wt, np.float32)
    hb = rne(v)
    r = v - f32(hb)
    mb = rne(r)
    lres = r - f32(mb)
    lb = rne(lres)
    assert np.array_equal(f32(lb), lres), 'the third level must be exact'
    levels = [x.astype(np.uint16) for x in (hb, mb, lb)]
    Q = D // 32
    out = np.zeros(head.npad * D * 3, np.uint16).reshape(head.npad // 16, Q, 3, 64, 8)
    for p, lev in enumerate(levels):
        # lev[col = 16 nt + c16][d = 32 tq + 8 kq + j] -> [nt][tq][p][lane = 16 kq + c16][j]
        out[:, :, p] = lev.reshape(head.npad // 16, 16, Q, 4, 8).transpose(0, 2, 3, 1, 4).reshape(head.npad // 16, Q, 64, 8)
    return out.reshape(-1)


X3 = {'CFL_DEBUG_PROJ_X3': '1'}      # the LDS-shared form (cfl_proj_x3_kernel) whatever the thresholds say
BX3 = {'CFL_DEBUG_PROJ_BX3': '1'}    # the chunk-at-a-time form (cfl_proj_bx3_kernel) also without kept planes
CASES = [
    # style, dist, D, L, K, B, loss kwargs, env (the forward form under test, in both runs)
    ('dist', 'pcd', 4096, 20, 3, 512, dict(), BX3),                                    # headline: half tiles, no hand-off
    ('dist', 'pcd', 4096, 20, 3, 1024, dict(), X3),                                    # half tiles, no row split
    ('dist', 'pcd', 4096, 20, 3, 1536, dict(reg_const=1e-3), X3),                      # half tiles, rows split in two
    ('cfl', 'pcd', 2048, 20, 5, 1024, dict(pos_weight=0.25), BX3),                     # config 4: 64-d tiles, P = 4, weight norm
    ('cfl', 'pcd', 2048, 20, 5, 1024, dict(pos_weight=0.25), X3),
    ('cfl', 'siamese', 1024, 256, 1, 512, dict(use_threshold=False, caffe_margin=100.0, pos_weight=0.0625), BX3),   # config 3
    ('cfl', 'siamese', 1024, 256, 1, 512, dict(use_threshold=False, caffe_margin=100.0, pos_weight=0.0625), X3),
    ('cfl', 'monomer', 2048, 32, 3, 1024, dict(), BX3),
    ('cfl', 'monomer', 2048, 32, 3, 1024, dict(), X3),
    ('dist', 'pcd', 1088, 7, 2, 100, dict(), BX3),                                     # D % 128 == 64: a half chunk at the end
]


@pytest.mark.parametrize('style,dist,D,L,K,B,lkw,env', CASES)
def test_kept_planes_equal_per_call_split(style, dist, D, L, K, B, lkw, env, monkeypatch):
    rng = np.random.RandomState(11)
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style=style)
    p = _params(cfg, rng)
    sh = H.make_shape(D, L, K, dist, cfg.weight_norm, cfg.has_bias)
    lay = H.layout(sh)
    norm, loss = H.make_norm(1.0 / 31.9098), H.make_loss(**lkw)
    pool = [[torch.from_numpy((np.abs(rng.randn(B, D)) * 8.0).astype(np.float32)).cuda() for _ in range(4)] for _ in range(3)]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    H.reload_env()

    def run(keep):
        theta = H.pack_theta(sh, p, None, 0.5 if dist != 'siamese' else 40.0, 'cuda')
        m, v, grad = torch.zeros_like(theta), torch.zeros_like(theta), torch.zeros_like(theta)
        scal = torch.zeros(H.S_COUNT, device='cuda')
        ws = torch.full((H.workspace_bytes(sh, B, 2) // 4,), float('nan'), dtype=torch.float32, device='cuda')
        planes = H.ThetaPlanes(sh, 'cuda') if keep else None
        if keep:
            planes.buf.fill_(0x7fc0)          # poison: bf16 NaN in every slot
        hist = []
        for i in range(12):
            H.pair_train_step(sh, norm, loss, pool[i % 3], theta, m, v, grad, scal, ws, 1e-3, 0.9, 0.999, planes=planes)
            hist.append(scal.clone())
            if keep:
                assert planes.valid, 'the fused step must leave the kept planes valid'
        torch.cuda.synchronize()
        return theta, m, v, grad, torch.stack(hist), planes
    try:
        a, b = run(False), run(True)
    finally:
        for k in env:
            monkeypatch.delenv(k)
        H.reload_env()
    assert not torch.isnan(a[0]).any() and float(a[4][:, H.S_ERROR].abs().max()) == 0.0
    for x, y, name in zip(a[:5], b[:5], ('theta', 'm', 'v', 'grad', 'scalars')):
        assert torch.equal(x, y), name
    # the kept planes are the split of the final theta, bit for bit, for every head a side projects through
    th = b[0].cpu().numpy()
    got = b[5].buf.cpu().numpy().view(np.uint16)
    enc = lay.enc[0]
    heads = {'pcd': (enc.proto, enc.outputs), 'monomer': (enc.outputs, enc.proto), 'siamese': (enc.outputs,)}[dist]
    for head in heads:
        want = _host_planes(th, head, D)
        assert np.array_equal(got[3 * head.w:3 * head.w + want.size], want), 'planes of the head at %d' % head.w


def test_kept_planes_are_resplit_after_an_external_update(monkeypatch):
    """the caller clears `valid` when theta changes behind the library's back: the next step splits again"""
    from cfl.engine import PairEngine
    rng = np.random.RandomState(5)
    D, L, K, B = 4096, 20, 3, 1024
    cfg = O.EncoderCfg(D=D, L=L, K=K)
    p = _params(cfg, rng)
    monkeypatch.setenv('CFL_DEBUG_PROJ_X3', '1')
    H.reload_env()
    try:
        batch = [torch.from_numpy((np.abs(rng.randn(B, D)) * 8.0).astype(np.float32)).cuda() for _ in range(4)]
        ea = PairEngine(D, L, K, norm=H.make_norm(1 / 31.9), params=p, batch_size=B)
        eb = PairEngine(D, L, K, norm=H.make_norm(1 / 31.9), params=p, batch_size=B)
        for _ in range(3):
            ea.step(batch)
            eb.step(batch)
        assert ea.planes.valid
        new = ea.theta.clone() * 1.01
        ea.set_theta(new)
        assert not ea.planes.valid
        eb.planes = None                      # reference run: no kept planes at all
        eb.theta.copy_(new)
        for _ in range(3):
            ea.step(batch)
            eb.planes = None
            H.pair_train_step(eb.shape, eb.norm, eb.loss, batch, eb.theta, eb.m, eb.v, eb.grad, eb.scalars,
                              eb._workspace(B, 2), eb.lr_t(), eb.beta1, eb.beta2, eb.eps)
            eb._advance()
        torch.cuda.synchronize()
        assert torch.equal(ea.theta, eb.theta)
    finally:
        monkeypatch.delenv('CFL_DEBUG_PROJ_X3')
        H.reload_env()


@pytest.mark.parametrize('style,dist,D,L,K,B,env', [
    ('dist', 'pcd', 1024, 20, 3, 256, {'CFL_DEBUG_P': '2'}),          # 64-d tiles, row split: publish / finish hand-off
    ('cfl', 'pcd', 4096, 20, 3, 512, {}),                             # weight norm: the finishers wait for the c_j sums
    ('cfl', 'siamese', 1024, 256, 1, 512, {}),                        # half tiles with the siamese pairing
])
def test_lost_hand_off_raises_instead_of_training_on_nan(style, dist, D, L, K, B, env, monkeypatch):
    """CFL_DEBUG_SPIN_LIMIT=-1 makes every finisher give up without polling: the step must report it through the
    sticky error word -- PairEngine.read_scalars raises CflHipError -- and a healthy engine must not."""
    from cfl.engine import PairEngine
    rng = np.random.RandomState(7)
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style=style)
    p = _params(cfg, rng)
    batch = [torch.from_numpy((np.abs(rng.randn(B, D)) * 8.0).astype(np.float32)).cuda() for _ in range(4)]

    def engine():
        return PairEngine(D, L, K, dist, weight_norm=cfg.weight_norm, has_bias=cfg.has_bias, norm=H.make_norm(1 / 31.9),
                          loss=H.make_loss(use_threshold=dist != 'siamese', caffe_margin=100.0 if dist == 'siamese' else None),
                          params=p, batch_size=B)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    H.reload_env()
    try:
        ok = engine()
        ok.step(batch)
        s = ok.read_scalars()
        assert np.isfinite(s['total'])
        monkeypatch.setenv('CFL_DEBUG_SPIN_LIMIT', '-1')
        H.reload_env()
        bad = engine()
        bad.step(batch)
        with pytest.raises(H.CflHipError, match='hand-off was lost'):
            bad.read_scalars()
        # sticky: a later healthy step does not clear it
        monkeypatch.delenv('CFL_DEBUG_SPIN_LIMIT')
        H.reload_env()
        bad.step(batch)
        with pytest.raises(H.CflHipError, match='hand-off was lost'):
            bad.read_scalars()
        raw = bad.scalars.cpu().numpy()
        assert H.lib().cfl_scalars_status(raw.ctypes.data) == -5
    finally:
        for k in list(env) + ['CFL_DEBUG_SPIN_LIMIT']:
            monkeypatch.delenv(k, raising=False)
        H.reload_env()


@pytest.mark.parametrize('style,dist,D,L,K,B,nv,lkw', [
    ('dist', 'pcd', 4096, 20, 3, 512, 58.388599, dict()),
    ('cfl', 'pcd', 2048, 20, 5, 1024, 1.0, dict(pos_weight=0.25)),
    ('cfl', 'siamese', 1024, 256, 1, 512, 31.9098, dict(use_threshold=False, caffe_margin=100.0, pos_weight=0.0625)),
    ('cfl', 'monomer', 1024, 64, 3, 128, 31.9098, dict(pos_weight=0.0625)),
    ('dist', 'pcd', 1088, 7, 2, 100, 1.0, dict(reg_const=1e-3)),
])
def test_bx3_forward_is_fp32_equivalent(style, dist, D, L, K, B, nv, lkw):
    """The training step with kept planes (cfl_proj_bx3_kernel: six round-to-nearest bf16 partial products per block)
    against the float64 oracle, beside the exact-fp32 forward of the same step: every scalar within 1e-5, every gradient
    tensor within the recorded-bar scale of tests/test_hip_parity.py (3e-6), and no error more than twice the
    exact-fp32 form's (floored at 1e-6 of the scale)."""
    rng = np.random.RandomState(4321)
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style=style)
    lcfg = O.LossCfg(**lkw)
    p = _params(cfg, rng)
    thr = 0.9 if dist != 'siamese' else 40.0
    batch = tuple((np.abs(rng.randn(B, D)) * nv / 4).astype(np.float32) for _ in range(4))
    b64 = tuple(b.astype(np.float64) / nv for b in batch)
    sc, g, _, dthr, dthr_aux = O.train_step_loss_and_grads(
        cfg, lcfg, {k: v.astype(np.float64) for k, v in p.items()}, np.float64(thr), b64, None)
    sh = H.make_shape(D, L, K, dist, cfg.weight_norm, cfg.has_bias)
    dev = [torch.from_numpy(b).cuda() for b in batch]

    def run(keep):
        theta = H.pack_theta(sh, p, None, thr, 'cuda')
        m, v = torch.zeros_like(theta), torch.zeros_like(theta)
        grad = torch.full_like(theta, float('nan'))
        scal = torch.zeros(H.S_COUNT, device='cuda')
        ws = torch.full((H.workspace_bytes(sh, B, 2) // 4,), float('nan'), dtype=torch.float32, device='cuda')
        planes = H.ThetaPlanes(sh, 'cuda') if keep else None
        H.pair_train_step(sh, H.make_norm(1.0 / nv), H.make_loss(**lkw), dev, theta, m, v, grad, scal, ws, 1e-3, 0.9, 0.999,
                          planes=planes)
        torch.cuda.synchronize()
        s = dict(zip(H.SCALAR_NAMES, scal.cpu().numpy().astype(np.float64)))
        gp, _, gthr = H.unpack_theta(sh, grad)
        return s, gp, gthr
    errs = {}
    for keep in (False, True):
        s, gp, gthr = run(keep)
        e = {}
        for k in ('total', 'thres', 'loss_pos', 'loss_neg', 'cd', 'accuracy', 'mean_d_pos', 'mean_d_neg'):
            e['s:' + k] = abs(s[k] - float(sc[k])) / max(1.0, abs(float(sc[k])))
            assert e['s:' + k] <= 1e-5, (keep, k, s[k], sc[k])
        gmax = max(float(np.abs(np.asarray(v)).max()) for v in g.values())
        for k in gp:
            r = np.asarray(g.get(k, np.zeros_like(gp[k])), dtype=np.float64)
            scale = max(np.abs(r).max(), 1e-3 * gmax, 1e-6)
            e['g:' + k] = float(np.abs(gp[k] - r).max() / scale)
            assert e['g:' + k] <= 3e-6, (keep, k, e['g:' + k])
        errs[keep] = e
    for k in errs[True]:
        assert errs[True][k] <= max(2.0 * errs[False][k], 1e-6), (k, errs[True][k], errs[False][k])
