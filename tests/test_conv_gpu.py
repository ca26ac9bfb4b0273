"""GPU parity of the convolution building blocks and of the head input-gradient against
the torch-CPU float64 oracle (oracle/conv_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import cfl_oracle as O
from oracle import conv_oracle as CO

pytestmark = pytest.mark.gpu

H = None


@pytest.fixture(scope='module', autouse=True)
def _hip():
    global H
    from cfl import hipabi
    hipabi.lib()
    H = hipabi
    yield


CONV_CASES = [
    # B, H, W, Ci, Co, K, stride, act, bias
    (6, 28, 28, 1, 64, 5, 2, 'lrelu', True),       # ConvPCD conv1 (Fashion-MNIST)
    (6, 14, 14, 64, 128, 5, 2, 'lrelu', True),     # ConvPCD conv2
    (3, 7, 9, 5, 20, 3, 1, None, False),           # odd sizes, residual-block style 3x3 s1
    (2, 8, 8, 6, 70, 4, 2, 'relu', True),          # 4x4 s2 (SRDiscriminator), even kernel -> asymmetric SAME pad
    (2, 5, 5, 3, 12, 3, 2, 'lrelu', True),         # odd input with stride 2
    (48, 14, 14, 64, 128, 5, 2, 'lrelu', True),    # training-size conv2: split-K weight gradient
    (48, 28, 28, 1, 64, 5, 2, 'lrelu', True),      # training-size conv1
    # 3x3 stride 1 with 32-channel multiples: the direct halo-tile kernel (csrc/conv_halo.h) for the forward pass
    # and the input gradient.  Tile variants: 8x16 pixels of one image / two 8x8 images / eight 4x4 images;
    # 128 / 64 / 32 output channels per workgroup; with and without the activation slope; split-K over channel
    # chunks (few output tiles); batch sizes that leave the last multi-image tile partly empty; channel counts
    # that are not a multiple of the channel tile.
    (3, 16, 16, 64, 128, 3, 1, 'lrelu', True),     # one image per tile, 128-channel tile both ways
    (2, 32, 32, 32, 32, 3, 1, 'lrelu', True),      # discriminator stage 1: 32 -> 32, 8 tiles per image
    (2, 16, 32, 64, 64, 3, 1, None, True),         # 64-channel tile, no slope, non-square image
    (5, 8, 8, 128, 128, 3, 1, 'lrelu', True),      # two images per tile, odd batch
    (11, 4, 4, 256, 256, 3, 1, 'relu', True),      # eight images per tile, partial last tile, split-K
    (3, 4, 4, 64, 512, 3, 1, None, False),         # generator block 1 shape: dx has N = 64, K = 512 (split-K)
    (2, 8, 8, 96, 160, 3, 1, 'lrelu', True),       # channels not multiples of the tiles (160 -> 2 x 128, 96 -> 128)
    (2, 16, 16, 32, 36, 3, 1, None, True),         # Co % 32 != 0: forward on the halo kernel, dx on the gathered GEMM
    (2, 4, 4, 256, 512, 3, 1, None, True),         # > 2^20 filter elements: row-wise (coalesced) scale / weight-norm finalisation
    # weight gradient on the halo-tile kernel: many pixel tiles per split and several splits (batch 40 of 32x32: 320 tiles),
    # relu slope, 3 x 2 channel blocks; an odd batch of 8x8 images with a ragged last two-image tile and several splits
    (40, 32, 32, 96, 64, 3, 1, 'relu', True),
    (37, 8, 8, 64, 96, 3, 1, 'lrelu', True),
    # fewer than 32 output channels on the halo kernel's 32-column tile (the generator's image layer: 128 -> 12): forward with
    # N = 12; input gradient with N = Ci = 12
    (2, 16, 16, 64, 12, 3, 1, None, True),
    (3, 8, 8, 12, 64, 3, 1, 'lrelu', True),
    # the 3-channel image stem (4x4 stride 2, csrc/conv_stem.h): the config-5 shape; a ragged 16-pixel segment without
    # activation / bias; 64 channels with two segments per row; the narrowest image (both edges in one patch row);
    # enough pixels for several weight-gradient slabs (pre-summed in groups)
    (3, 64, 64, 3, 32, 4, 2, 'lrelu', True),
    (2, 16, 24, 3, 32, 4, 2, None, False),
    (5, 8, 40, 3, 64, 4, 2, 'relu', True),
    (2, 6, 4, 3, 16, 4, 2, 'lrelu', True),
    (40, 32, 32, 3, 32, 4, 2, 'lrelu', True),
]


@pytest.mark.parametrize('B,Hh,Ww,Ci,Co,K,S,act,bias', CONV_CASES)
def test_conv2d_wn_fwd_bwd(B, Hh, Ww, Ci, Co, K, S, act, bias):
    rng = np.random.RandomState(11)
    x = rng.randn(B, Hh, Ww, Ci)
    V = rng.randn(K, K, Ci, Co) * 0.2
    g = 1.0 + 0.3 * rng.randn(Co)
    b = 0.1 * rng.randn(Co) if bias else None
    reg = 1e-3
    tx, tV, tg = (torch.tensor(a, requires_grad=True) for a in (x, V, g))
    tb = torch.tensor(b, requires_grad=True) if bias else None
    y = CO.conv2d_weight_norm(tx, tV, tg, tb, S, act)
    dy = torch.tensor(rng.randn(*y.shape))
    ((y * dy).sum() + 0.5 * reg * (tV * tV).sum()).backward()

    conv = H.make_conv(B, Hh, Ww, Ci, Co, K, K, S, act)
    direct = K == 3 and S == 1 and ((Ww % 16 == 0 and Hh % 8 == 0) or (Hh, Ww) in ((4, 4), (8, 8)))
    assert H.conv_uses_direct_kernel(conv, 'fwd') == (direct and Ci % 32 == 0 and Co % 4 == 0)
    assert H.conv_uses_direct_kernel(conv, 'dx') == (direct and Co % 32 == 0 and Ci % 4 == 0)
    # weight gradient on the halo-tile kernel (csrc/conv_halo_wgrad.h): 32-channel multiples on both sides, not the 4x4 images
    assert H.conv_uses_direct_kernel(conv, 'dw') == (direct and (Hh, Ww) != (4, 4) and Ci % 32 == 0 and Co % 32 == 0)
    assert H.conv_uses_direct_kernel(conv, 'stem') == (K == 4 and S == 2 and Ci == 3 and Co in (16, 32, 64)
                                                       and Hh % 2 == 0 and Ww % 2 == 0 and Ww >= 4)
    ws = H.conv_workspace(conv, 'cuda')
    f = lambda a: torch.tensor(a, dtype=torch.float32, device='cuda').contiguous()
    dx_, dV_, dg_, db_ = None, None, None, None
    gy = H.conv2d_wn_fwd(conv, f(x), f(V), f(g), f(b) if bias else None, ws)
    assert gy.shape == y.shape
    scale = max(1.0, float(y.abs().max()))
    assert float((gy.cpu().double() - y.detach()).abs().max()) <= 1e-5 * scale
    dx_, dV_, dg_, db_ = H.conv2d_wn_bwd(conv, f(x), f(V), f(g), gy, f(dy.numpy()), ws, reg_const=reg,
                                         need_db=bias)
    for got, ref, name in ((dx_, tx.grad, 'dx'), (dV_, tV.grad, 'dV'), (dg_, tg.grad, 'dg'),
                           (db_, tb.grad if bias else None, 'db')):
        if ref is None:
            continue
        err = float((got.cpu().double() - ref).abs().max())
        assert err <= 2e-5 * max(1e-2, float(ref.abs().max())), (name, err, float(ref.abs().max()))


def test_same_padding_and_subpixel_kats():
    """TF 'SAME' for even kernels pads the extra pixel on the bottom/right; the oracle's
    depth_to_space is block-major."""
    assert CO.same_pads(64, 4, 2) == (32, 1, 1)
    assert CO.same_pads(28, 5, 2) == (14, 1, 2)
    assert CO.same_pads(7, 3, 1) == (7, 1, 1)
    x = torch.arange(2 * 2 * 8, dtype=torch.float64).reshape(1, 2, 2, 8)
    y = CO.conv2d_subpixel(x, 2)
    for h in range(2):
        for w in range(2):
            for i in range(2):
                for j in range(2):
                    for c in range(2):
                        assert y[0, 2 * h + i, 2 * w + j, c] == x[0, h, w, (i * 2 + j) * 2 + c]


@pytest.mark.parametrize('style,dist,K,L', [('cfl', 'pcd', 3, 10), ('cfl', 'monomer', 2, 12),
                                            ('cfl', 'siamese', 1, 24), ('dist', 'pcd', 2, 6)])
def test_pair_input_grad(style, dist, K, L):
    """dL/dx of the pair step (needed by ConvPCD) against the oracle's fc_head_bwd dx."""
    rng = np.random.RandomState(8)
    D, B = 128, 20
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist, style=style)
    lcfg = O.LossCfg(pos_weight=0.5, lambda_m=0.25)
    p = O.init_encoder_params(cfg, rng, np.float64)
    for k in p:
        p[k] = p[k] + 0.05 * rng.randn(*p[k].shape)
    batch = tuple(rng.randn(B, D) for _ in range(4))
    thr = 0.7
    tb = [torch.tensor(b, requires_grad=True) for b in batch]
    tp = {k: torch.tensor(v) for k, v in p.items()}
    import tests.test_oracle as TO
    total, _, _ = TO._torch_forward(cfg, lcfg, tp, torch.tensor(thr), tuple(tb))
    total.backward()
    sh = H.make_shape(D, L, K, dist, cfg.weight_norm, cfg.has_bias)
    theta = H.pack_theta(sh, {k: v.astype(np.float32) for k, v in p.items()}, None, thr, 'cuda')
    grad = torch.empty_like(theta)
    scal = torch.zeros(H.S_COUNT, device='cuda')
    ws = torch.empty(H.workspace_bytes(sh, B, 2) // 4, dtype=torch.float32, device='cuda')
    norm = H.make_norm()
    dev = [torch.tensor(b, dtype=torch.float32, device='cuda') for b in batch]
    H.pair_step_fwd_bwd(sh, norm, H.make_loss(pos_weight=0.5, lambda_m=0.25), dev, theta, grad, scal, ws)
    dsrc, ddst = H.pair_input_grad(sh, norm, B, theta, ws)
    ref_src = torch.cat([tb[0].grad, tb[2].grad])
    ref_dst = torch.cat([tb[1].grad, tb[3].grad])
    for got, ref in ((dsrc, ref_src), (ddst, ref_dst)):
        err = float((got.cpu().double() - ref).abs().max())
        assert err <= 2e-5 * max(1e-3, float(ref.abs().max())), err


@pytest.mark.parametrize('B', [12, 100])
def test_convpcd_model_matches_oracle(B):
    """BASELINE config 0 shape (Fashion-MNIST 28x28x1, conv encoder, PCD K=1, latent 30,
    sigmoid data, reg 5e-4; B = 100 is the reference's batch size): CFL --model-type conv against the torch-autograd
    float64 oracle (conv trunk of oracle/conv_oracle.py + heads/distance/loss of tests/test_oracle.py +
    TF-Adam of oracle/cfl_oracle.py), 6 training steps on identical batches -- beside the SAME oracle evaluated in float32
    on the CPU: after step 0 the loss must be no further from float64 than twice the fp32 CPU evaluation is
    (tests/parity_series.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity_series import ParitySeries
    import tests.test_oracle as TO
    from cfl import ops
    from cfl.models.cfl import construct_model
    rng = np.random.RandomState(4)
    shape, L, K, reg = (28, 28, 1), 30, 1, 5e-4
    dn = ops.dist_normalizer(shape, None, None, None, None, None, 'sigmoid')
    kw = dict(is_double=False, disable_double=False, latent_shape=None, source_shape=None, input_shape=shape,
              ae_shape=None, batch_size=B, data_norm=None, data_type='sigmoid', model_type='conv',
              gan_type='conv', num_components=K, latent_size=L, pos_weight=None, caffe_margin=None, gan=False,
              cgan=False, t_dim=None, dist_type='pcd', act_type=None, use_threshold=True, lr=1e-3, beta1=0.9,
              beta2=0.999, z_dim=20, z_stddev=1., g_dim=64, g_lr=2e-4, g_beta1=.5, g_beta2=.999, m_prj=None,
              m_enc=None, d_dim=64, d_lr=2e-4, d_beta1=.5, d_beta2=.999, lambda_dra=.5, lambda_gp=None,
              lambda_m=0.0, directed=False, data_directed=False, reg_const=reg, data_normalizer=dn[0],
              data_unnormalizer=dn[1], seed=2)
    model, _ = construct_model(**kw)
    assert model.get_name() == 'cfl_pcd_conv_sigmoid_ls_30_nc_1_ut_reg_0.0005'
    assert model.trunk.feature_size == 6272

    # oracle parameters = the model's initial variables
    model.engine.theta[model.engine.layout.thr] = 0.3   # off the max(thr, 1e-6) tie: torch.maximum
    hp, _, thr = model.engine.named_variables()         # splits the gradient there, TF does not
    cfg = O.EncoderCfg(D=6272, L=L, K=K, dist_type='pcd', style='cfl')
    params = {'head/' + k: v.astype(np.float64) for k, v in hp.items()}
    for k, v in model.trunk.named().items():
        params['conv/' + k] = v.astype(np.float64)
    params['thr'] = np.float64(thr)
    adam, adam32 = O.AdamState(1e-3), O.AdamState(1e-3)
    params32 = {k: np.asarray(v, np.float32) for k, v in params.items()}
    lcfg = O.LossCfg(reg_const=reg)
    ser = ParitySeries('convpcd_config0_b%d' % B, floor=2e-5, meta=dict(shape=shape, B=B, steps=6, L=L, K=K, reg=reg))
    # the STRICT comparison: a second float64 oracle is stepped beside the free-running one with the HIP
    # run's lrelu slope patterns imposed (tests/test_activation_masks_gpu.py) -- it differs from the HIP trajectory by rounding
    # only, not by kink flips, and the variables are held against IT entry by entry below
    from test_activation_masks_gpu import MaskedConvRun
    masked = MaskedConvRun(model, hp, thr, cfg, lcfg, shape, reg, B, np.float64)
    masked_err = []

    def oracle_loss(p, batch, dtype=torch.float64):
        tp = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in p.items()}
        cp = {k.split('/', 1)[1].replace('/Conv/', '/').replace('biases', 'b'): v
              for k, v in tp.items() if k.startswith('conv/')}
        feats = [CO.convpcd_features(torch.clamp(torch.tensor(b, dtype=dtype), 0., 1.), shape, cp)
                 for b in batch]
        head = {k.split('/', 1)[1]: v for k, v in tp.items() if k.startswith('head/')}
        total, _, _ = TO._torch_forward(cfg, lcfg, head, tp['thr'], tuple(feats))
        total = total + sum(0.5 * reg * (v * v).sum() for k, v in cp.items() if k.endswith('/V'))
        total.backward()
        return float(total), {k: (v.grad.numpy() if v.grad is not None else np.zeros_like(p[k]))
                              for k, v in tp.items()}

    for step in range(6):
        batch = tuple(rng.rand(B, 784).astype(np.float32) * 1.2 - 0.1 for _ in range(4))   # exercises the clip
        ref, grads = oracle_loss(params, batch)
        ref32, grads32 = oracle_loss(params32, batch, torch.float32)
        model.train_step(batch)
        got = model.scalars()['total']
        acts = [a.clone() for a in model.trunk._inputs[1:]]  # post-lrelu, rows [pos_src | neg_src | pos_dst | neg_dst]
        ref_masked = masked.step(batch, acts)
        masked_err.append(abs(got - ref_masked) / max(1.0, abs(ref_masked)))
        # fp32-vs-fp64 sign flips of near-zero lrelu pre-activations change single slopes (0.2 <-> 1), for the fp32 CPU
        # evaluation exactly as here: the free-running trajectory is held to twice the fp32 CPU twin's distance from float64
        # (tests/test_activation_masks_gpu.py: with the masks held equal the same steps agree to < 1e-4)
        ser.add(step, 'total', got, ref, ref32)
        # ... and, while the fp32 CPU evaluation itself is still within north_star's 1e-5 of float64 (no kink has flipped yet),
        # within 1e-5 of IT: the reference's arithmetic is an fp32 CPU path.  (Observed on the recording box: <= 1e-6 at
        # every step, also after both have left float64 by 2e-4 .. 5e-4 -- profiles/r05_series/series_convpcd_config0_*.json --
        # but which kinks flip depends on the host's BLAS threading, so the late steps are not asserted.)
        if abs(ref32 - ref) <= 1e-5 * max(1.0, abs(ref)):
            assert abs(got - ref32) <= 1e-5 * max(1.0, abs(ref)), (step, got, ref32, ref)
        if step == 0:
            assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref)), (step, got, ref)
        adam.apply(params, grads)
        adam32.apply(params32, {k: np.asarray(v, np.float32) for k, v in grads32.items()})
    ser.check()
    # variables after 6 Adam steps
    hp, _, thr = model.engine.named_variables()
    # STRICT: against the equal-mask float64 oracle -- losses of every step within 1e-4 (fp32 rounding amplified by Adam's lr-sized
    # first steps; the free-running bar is 5e-4), and >= 99 % of the entries of EVERY variable within 2e-5 of the tensor's scale
    # (the rest: entries whose gradient sits at the fp32 noise level, which Adam moves by ~lr either way), none further than
    # one Adam step per iteration
    assert max(masked_err) <= 1e-4 and masked_err[0] <= 2e-6, ['%.1e' % e for e in masked_err]
    for got_v, ref_v, k in [(v, masked.params['head/' + k], k) for k, v in hp.items()] + \
                           [(v, masked.params['conv/' + k], k) for k, v in model.trunk.named().items()]:
        d = np.abs(got_v - ref_v)
        assert (d <= 2e-5 * max(1.0, np.abs(ref_v).max())).mean() >= 0.99, (k, float(d.max()), float((d <= 2e-5).mean()))
        assert d.max() <= 6 * 1e-3 * 1.05, (k, float(d.max()))
    # free-running SANITY bound (the oracle without the masks; kink flips included):
    # Adam's first steps move every weight by ~lr * sign(g): one flipped lrelu slope changes the
    # sign of a few tiny gradients of one channel, hence a robust comparison (>= 97 % of the
    # entries agree to 1e-4 of the tensor scale; nothing is off by more than 2 * steps * lr)
    def close(v, ref, k):
        diff, scale = np.abs(v - ref), max(1e-2, np.abs(ref).max())
        tol = max(1e-4 * scale, 0.05 * 6 * 1e-3)     # 5 % of the distance Adam can travel
        assert np.mean(diff <= tol) >= 0.97, (k, np.mean(diff <= tol))
        assert diff.max() <= 2 * 6 * 1e-3, (k, diff.max())
    for k, v in hp.items():
        close(v, params['head/' + k], k)
    for k, v in model.trunk.named().items():
        close(v, params['conv/' + k], k)
    st = model.checkpoint_state()['variables']
    assert st['CFL/DistEncoder/conv2/Conv/V'].shape == (5, 5, 64, 128)
    assert st['CFL/DistEncoder/outputs/fully_connected/V'].shape == (6272, 30)


def test_fcpcd_hidden_layers_match_oracle():
    """FCPCD(layer_sizes=[...]) -- hidden weight-normalised fc_i + lrelu layers in front of the heads
    (cfl/models/blocks.py:509-524; no command line of the reference sets them, the model classes take the argument).  CFL linear
    model, D = 256, layer_sizes [128, 64], pcd K = 3, L = 10, reg 5e-4 (V of the hidden layers regularised, their biases not),
    4 training steps on identical batches against the float64 torch-autograd oracle (oracle/conv_oracle.py::fcpcd_hidden + heads /
    distance / loss of tests/test_oracle.py + TF-Adam): loss of step 0 within 2e-5, every step within twice the fp32 CPU
    evaluation's distance (lrelu kinks), variables after the steps, and the checkpoint names / shapes of SURVEY App. D."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity_series import ParitySeries
    import tests.test_oracle as TO
    from cfl import ops
    from cfl.models.cfl import construct_model
    rng = np.random.RandomState(8)
    D, L, K, B, reg, sizes, steps = 256, 10, 3, 16, 5e-4, [128, 64], 4
    dn = ops.dist_normalizer((D,), None, None, None, [4.0], None, 'linear')
    kw = dict(is_double=False, disable_double=False, latent_shape=None, source_shape=None, input_shape=(D,),
              ae_shape=None, batch_size=B, data_norm=[4.0], data_type='linear', model_type='linear',
              gan_type='conv', num_components=K, latent_size=L, pos_weight=0.25, caffe_margin=None, gan=False,
              cgan=False, t_dim=None, dist_type='pcd', act_type=None, use_threshold=True, lr=1e-3, beta1=0.9,
              beta2=0.999, z_dim=20, z_stddev=1., g_dim=64, g_lr=2e-4, g_beta1=.5, g_beta2=.999, m_prj=None,
              m_enc=None, d_dim=64, d_lr=2e-4, d_beta1=.5, d_beta2=.999, lambda_dra=.5, lambda_gp=None,
              lambda_m=0.0, directed=False, data_directed=False, reg_const=reg, data_normalizer=dn[0],
              data_unnormalizer=dn[1], seed=2, layer_sizes=sizes)
    model, _ = construct_model(**kw)
    assert model.trunk is not None and model.trunk.feature_size == 64 and model.engine.shape.D == 64
    model.engine.theta[model.engine.layout.thr] = 0.3
    hp, _, thr = model.engine.named_variables()
    cfg = O.EncoderCfg(D=64, L=L, K=K, dist_type='pcd', style='cfl')
    lcfg = O.LossCfg(reg_const=reg, pos_weight=0.25)
    params = {'head/' + k: v.astype(np.float64) for k, v in hp.items()}
    for k, v in model.trunk.named().items():
        params['fc/' + k] = v.astype(np.float64)
    params['thr'] = np.float64(thr)
    params32 = {k: np.asarray(v, np.float32) for k, v in params.items()}
    adam, adam32 = O.AdamState(1e-3), O.AdamState(1e-3)
    ser = ParitySeries('fcpcd_hidden_layers', floor=2e-5, meta=dict(D=D, B=B, steps=steps, L=L, K=K, reg=reg, layer_sizes=sizes))

    def oracle_loss(p, batch, dtype=torch.float64):
        tp = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in p.items()}
        fp = {k.split('/', 1)[1].replace('/fully_connected/', '/').replace('biases', 'b'): v
              for k, v in tp.items() if k.startswith('fc/')}
        feats = [CO.fcpcd_hidden(torch.tensor(b, dtype=dtype) / 4.0, fp) for b in batch]
        head = {k.split('/', 1)[1]: v for k, v in tp.items() if k.startswith('head/')}
        total, _, _ = TO._torch_forward(cfg, lcfg, head, tp['thr'], tuple(feats))
        total = total + sum(0.5 * reg * (v * v).sum() for k, v in fp.items() if k.endswith('/V'))
        total.backward()
        return float(total), {k: (v.grad.numpy() if v.grad is not None else np.zeros_like(p[k])) for k, v in tp.items()}

    for step in range(steps):
        batch = tuple((rng.randn(B, D) * 2.0).astype(np.float32) for _ in range(4))
        ref, grads = oracle_loss(params, batch)
        ref32, grads32 = oracle_loss(params32, batch, torch.float32)
        model.train_step(batch)
        got = model.scalars()['total']
        ser.add(step, 'total', got, ref, ref32)
        if step == 0:
            assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref)), (got, ref)
        adam.apply(params, grads)
        adam32.apply(params32, {k: np.asarray(v, np.float32) for k, v in grads32.items()})
    ser.check()
    hp, _, _ = model.engine.named_variables()
    for got_v, ref_v, k in [(v, params['head/' + k], k) for k, v in hp.items()] + \
                           [(v, params['fc/' + k], k) for k, v in model.trunk.named().items()]:
        d = np.abs(got_v - ref_v)
        tol = max(1e-4 * max(1e-2, np.abs(ref_v).max()), 0.05 * steps * 1e-3)
        assert np.mean(d <= tol) >= 0.97 and d.max() <= 2 * steps * 1e-3, (k, float(d.max()), float(np.mean(d <= tol)))
    st = model.checkpoint_state()['variables']
    assert st['CFL/DistEncoder/fc_0/fully_connected/V'].shape == (256, 128)
    assert st['CFL/DistEncoder/fc_1/fully_connected/V'].shape == (128, 64)
    assert st['CFL/DistEncoder/fc_1/fully_connected/biases'].shape == (64,)
    assert st['CFL/DistEncoder/outputs/fully_connected/V'].shape == (64, 10)
    # a checkpoint round trip restores the hidden layers too
    st_all = model.checkpoint_state()
    model.trunk.theta.zero_()
    model.load_checkpoint_state(st_all)
    assert np.array_equal(model.trunk.named()['fc_0/fully_connected/V'], st['CFL/DistEncoder/fc_0/fully_connected/V'])


def test_exact_fp32_gemm_path_in_a_fresh_process():
    """The conv GEMMs default to the bf16x3 matrix-core arithmetic; CFL_EXACT_FP32=1 (read once per process)
    selects the fp32-MFMA kernel.  Run two layer cases under it so that both cores stay covered."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, CFL_EXACT_FP32='1')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu',
                        os.path.join(root, 'tests', 'test_conv_gpu.py'), '-k',
                        'test_conv2d_wn_fwd_bwd and (6-14-14-64-128 or 2-8-8-6-70)'],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'deselected' in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize('B,Hh,Ww,Ci,Co,K,S,act', [
    (3, 16, 16, 64, 128, 3, 1, 'lrelu'),     # halo kernel, one image per tile
    (5, 8, 8, 128, 256, 3, 1, 'relu'),       # two images per tile; Co / 4 = 64
    (11, 4, 4, 256, 256, 3, 1, 'lrelu'),     # eight images per tile, split-K over channel chunks (the reduce kernel's epilogue)
    (2, 16, 16, 64, 12, 3, 1, None),         # the generator's output layer: Co / 4 = 3 -> halo kernel on a padded 32-column tile (round 5)
    (2, 8, 8, 6, 72, 4, 2, 'relu'),          # stride 2 / even kernel on the gathered GEMM
    (2, 16, 16, 32, 36, 3, 1, 'lrelu'),      # Co / 4 = 9 on the halo kernel: four consecutive channels straddle two quarters
])
def test_fused_store_epilogues_equal_the_separate_kernels(B, Hh, Ww, Ci, Co, K, S, act):
    """cfl_conv2d_wn_fwd_fused: y = act(conv + b + residual) must equal the convolution followed by cfl_ew_add_act, and the
    sub-pixel shuffled store must equal the convolution followed by cfl_subpixel2x_fwd -- BIT FOR BIT (same operations in
    the same order; the shuffle is a pure index remap), on the halo kernel, its split-K reduce kernel and the gathered GEMM."""
    from cfl import hipgan as G
    rng = np.random.RandomState(3)
    dev = 'cuda'
    x = torch.tensor(rng.randn(B, Hh, Ww, Ci).astype(np.float32), device=dev)
    V = torch.tensor((rng.randn(K, K, Ci, Co) * 0.2).astype(np.float32), device=dev)
    g = torch.tensor((1.0 + 0.3 * rng.randn(Co)).astype(np.float32), device=dev)
    b = torch.tensor((0.1 * rng.randn(Co)).astype(np.float32), device=dev)
    plain = H.make_conv(B, Hh, Ww, Ci, Co, K, K, S, None)
    fused = H.make_conv(B, Hh, Ww, Ci, Co, K, K, S, act)
    oh, ow = H.conv_out_hw(plain)
    ws = H.conv_workspace(plain, dev)
    y0 = torch.empty(B, oh, ow, Co, device=dev)
    G.conv_fwd(plain, x, V, g, b, y0, ws)
    res = torch.tensor(rng.randn(B, oh, ow, Co).astype(np.float32), device=dev)
    # residual join
    want = G.add_act(y0, res, act)
    got = torch.full_like(y0, float('nan'))
    G.conv_fwd(fused, x, V, g, b, got, ws, residual=res)
    assert torch.equal(got, want)
    # sub-pixel shuffled store (+ activation)
    want = G.subpixel_fwd(y0, act)
    got = torch.full((B, 2 * oh, 2 * ow, Co // 4), float('nan'), device=dev)
    G.conv_fwd(fused, x, V, g, b, got, ws, subpixel=True)
    assert torch.equal(got, want)
    # both, through a per-layer cache (the path the MrCGAN stacks take)
    cache = G.ConvCache()
    want = G.subpixel_fwd(G.add_act(y0, res, None), act)
    got = torch.full((B, 2 * oh, 2 * ow, Co // 4), float('nan'), device=dev)
    for _ in range(2):          # second call: cached scale / planes
        G.conv_fwd(fused, x, V, g, b, got, ws, cache=cache, residual=res, subpixel=True)
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize('B,Hh,Ww,Ci,Co,act', [
    (5, 8, 8, 64, 256, 'relu'),        # two-image tiles, ragged last tile, generator block shape in small
    (3, 16, 16, 32, 128, 'relu'),      # one image per tile: one 32-channel chunk per quarter
    (2, 16, 32, 64, 384, None),        # no activation after the shuffle (y not needed), three 32-channel chunks per quarter
    (4, 8, 8, 32, 128, 'lrelu'),
])
def test_backward_with_the_unshuffle_in_the_dy_loaders_equals_the_separate_kernel(B, Hh, Ww, Ci, Co, act):
    """cfl_conv2d_wn_bwd_fused(dy_subpixel = 1): dy and the activated output arrive 2x sub-pixel shuffled, the layout
    cfl_conv2d_wn_fwd_fused(subpixel = 1) stores; dx, dV, dg, db must equal cfl_subpixel2x_bwd followed by the plain backward
    BIT FOR BIT (the loaders form the same dy * act'(y) products and feed the same kernels), with and without the cache."""
    from cfl import hipgan as G
    rng = np.random.RandomState(5)
    dev = 'cuda'
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
    x, V = t(rng.randn(B, Hh, Ww, Ci)), t(rng.randn(3, 3, Ci, Co) * 0.2)
    g = t(1.0 + 0.3 * rng.randn(Co))
    s = t(rng.randn(B, 2 * Hh, 2 * Ww, Co // 4))          # stands for the activated, shuffled output (only its signs matter)
    if act == 'relu':
        s = torch.relu(s)
    d = t(rng.randn(B, 2 * Hh, 2 * Ww, Co // 4))
    plain = H.make_conv(B, Hh, Ww, Ci, Co, 3, 3, 1, None)
    fused = H.make_conv(B, Hh, Ww, Ci, Co, 3, 3, 1, act)
    assert G.conv_bwd_takes_subpixel(fused)
    ws = H.conv_workspace(plain, dev)
    outs = {}
    for mode in ('separate', 'fused', 'fused+cache'):
        dx = torch.full((B, Hh, Ww, Ci), float('nan'), device=dev)
        dV, dg, db = torch.full_like(V, float('nan')), torch.full_like(g, float('nan')), torch.full_like(g, float('nan'))
        if mode == 'separate':
            dyp = G.subpixel_bwd(s if act else None, d, act)
            G.conv_bwd(plain, x, V, g, None, dyp, ws, dx=dx, dV=dV, dg=dg, db=db, reg_const=1e-3)
        else:
            cache = G.ConvCache() if mode.endswith('cache') else None
            for _ in range(2 if cache is not None else 1):
                G.conv_bwd(fused, x, V, g, s if act else None, d, ws, dx=dx, dV=dV, dg=dg, db=db, reg_const=1e-3, cache=cache,
                           dy_subpixel=True)
        outs[mode] = (dx, dV, dg, db)
    for mode in ('fused', 'fused+cache'):
        for got, want, name in zip(outs[mode], outs['separate'], ('dx', 'dV', 'dg', 'db')):
            assert torch.equal(got, want), (mode, name, float((got - want).abs().max()))
    # shapes that cannot take the shuffled layout say so
    assert not G.conv_bwd_takes_subpixel(H.make_conv(B, 4, 4, Ci, Co, 3, 3, 1, act))
    assert not G.conv_bwd_takes_subpixel(H.make_conv(B, Hh, Ww, Ci, 96, 3, 3, 1, act))
    with pytest.raises(H.CflHipError):
        G.conv_bwd(H.make_conv(B, Hh, Ww, Ci, 96, 3, 3, 1, act), x, t(rng.randn(3, 3, Ci, 96)), t(np.ones(96)), None,
                   t(rng.randn(B, 2 * Hh, 2 * Ww, 24)), ws, dx=torch.empty_like(x), dy_subpixel=True)
