"""CPU oracle for the MrCGAN post-epoch step (TEST INFRASTRUCTURE ONLY).

torch (CPU, float64, autograd incl. double backward) restatement of

    SRGenerator               cfl/models/blocks.py:25-109
    SRDiscriminator           cfl/models/blocks.py:112-247
    ConvTransposeGenerator    cfl/models/blocks.py:250-332
    ConvDiscriminator         cfl/models/blocks.py:335-438
    GAN graph (non-cgan)      cfl/models/cfl.py:730-806
    GAN losses                cfl/models/cfl.py:951-1063
    the two Adams             cfl/models/cfl.py:1087-1096, run together at cfl/models/cfl.py:1491-1497

PARITY: composition pinned, TensorFlow kernels unpinned (see oracle/cfl_oracle.py).  The whole post-epoch
step -- generator / discriminator stacks, the GAN wiring of CFL._build_model, every loss part, the gradient
penalty's double backward, the two Adams, for srgan / conv, cgan / non-cgan, with and without --t-dim -- is
checked to float64 round-off against golden vectors produced by the reference's own classes running over a
TF stand-in (tests/golden/make_arith_goldens.py, tests/test_arith_goldens.py).  TF-1 semantics the stand-in
and this file both take from TF's documentation: conv2d_transpose with 'SAME' padding is
the exact adjoint of the 'SAME' strided convolution with the same filter; relu'(0) = 0; tf.nn.moments =
population variance.  Only tests/ and smoke() may import this file.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import conv_oracle as CO


def _up_count(shape, min_dim=4):
    start = min(shape[0], shape[1])
    n = 0
    while start % 2 == 0 and start > min_dim:
        start //= 2
        n += 1
    return n, start


def fc_wn(x, V, g, b, act=None):
    """cfl/layers.py:80-94 (no epsilon in the FC variant)."""
    y = (x @ V) * (g / torch.sqrt((V * V).sum(0)))
    if b is not None:
        y = y + b
    if act == 'lrelu':
        y = CO.lrelu(y)
    elif act == 'relu':
        y = CO.relu(y)
    return y


def conv2d_transpose_weight_norm(x, V, g, b, stride, activation=None):
    """x [B,H,W,Ci]; V [KH,KW,Co,Ci]; out [B,H*s,W*s,Co] (cfl/layers.py:253-361).
    Computed as the vector-Jacobian product of the forward 'SAME' convolution (which is
    how TF defines conv2d_transpose)."""
    KH, KW, Co, Ci = V.shape
    n2 = (V * V).sum(dim=(0, 1, 3), keepdim=True)
    W = V * torch.rsqrt(torch.clamp(n2, min=1e-12)) * g.reshape(1, 1, -1, 1)
    B, H, Wd, _ = x.shape
    OH, OW = H * stride, Wd * stride
    probe = torch.zeros(B, OH, OW, Co, dtype=x.dtype, requires_grad=True)
    _, pt, pb = CO.same_pads(OH, KH, stride)
    _, pl, pr = CO.same_pads(OW, KW, stride)
    fwd = F.conv2d(F.pad(probe.permute(0, 3, 1, 2), (pl, pr, pt, pb)), W.permute(3, 2, 0, 1), stride=stride)
    y, = torch.autograd.grad(fwd, probe, x.permute(0, 3, 1, 2), create_graph=True)
    if b is not None:
        y = y + b
    if activation == 'relu':
        y = CO.relu(y)
    return y


# ---- parameter construction (Xavier-uniform, g = 1, b = 0) ---------------------------------
def _conv_p(p, name, rng, kh, kw, ci, co, bias=True):
    p[name + '/V'] = CO.xavier(rng, (kh, kw, ci, co), kh * kw * ci, kh * kw * co)
    p[name + '/g'] = np.ones(co)
    if bias:
        p[name + '/biases'] = np.zeros(co)


def _fc_p(p, name, rng, d, n, bias=True):
    p[name + '/V'] = CO.xavier(rng, (d, n), d, n)
    p[name + '/g'] = np.ones(n)
    if bias:
        p[name + '/biases'] = np.zeros(n)


def _gen_in(p, rng, in_dim, c_dim, t_dim):
    """cgan with --t-dim: the condition first goes through fc_t (cfl/models/blocks.py:55-64); returns fc1's fan-in."""
    if t_dim:
        _fc_p(p, 'fc_t/fully_connected', rng, c_dim, t_dim)
        return in_dim - c_dim + t_dim
    return in_dim


def _gen_cat(zc, p, c_dim):
    if 'fc_t/fully_connected/V' in p:
        z, c = zc[:, :zc.shape[1] - c_dim], zc[:, zc.shape[1] - c_dim:]
        n = 'fc_t/fully_connected'
        c = fc_wn(c, p[n + '/V'], p[n + '/g'], p[n + '/biases'], 'lrelu')
        return torch.cat([z, c], 1)
    return zc


def init_sr_generator(ae_shape, in_dim, rng, dim=64, c_dim=None, t_dim=None):
    nb, start = _up_count(ae_shape)
    p = {}
    in_dim = _gen_in(p, rng, in_dim, c_dim, t_dim)
    _fc_p(p, 'fc1/fully_connected', rng, in_dim, dim * start * start)
    ci = dim
    for i in range(nb - 1):
        co = 4 * dim * (2 ** (nb - i - 1))
        _conv_p(p, 'subpixel_block%d/Conv' % (i + 1), rng, 3, 3, ci, co)
        ci = co // 4
    _conv_p(p, 'outputs/Conv', rng, 3, 3, ci, 4 * ae_shape[2])
    return p


def sr_generator(zc, p, ae_shape, data_type, dim=64, c_dim=None):
    nb, start = _up_count(ae_shape)
    zc = _gen_cat(zc, p, c_dim)
    h = fc_wn(zc, p['fc1/fully_connected/V'], p['fc1/fully_connected/g'], p['fc1/fully_connected/biases'], 'relu')
    h = h.reshape(-1, start, start, dim)
    for i in range(nb - 1):
        n = 'subpixel_block%d/Conv' % (i + 1)
        h = CO.conv2d_weight_norm(h, p[n + '/V'], p[n + '/g'], p[n + '/biases'], 1, None)
        h = CO.conv2d_subpixel(h, 2, 'relu')
    h = CO.conv2d_weight_norm(h, p['outputs/Conv/V'], p['outputs/Conv/g'], p['outputs/Conv/biases'], 1, None)
    h = CO.conv2d_subpixel(h, 2, None)
    out = h.reshape(h.shape[0], -1)
    return _data_act(out, data_type)


def _data_act(x, data_type):
    return {'linear': lambda t: t, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid, 'relu': torch.relu}[data_type](x)


def _cname(i):
    return 'Conv' if i == 0 else 'Conv_%d' % i


def _cond(p, scope, rng, c_dim, t_dim):
    """channels the tiled condition adds; with --t-dim it first goes through <scope>/fc_t"""
    if t_dim:
        _fc_p(p, scope + 'fc_t/fully_connected', rng, c_dim, t_dim)
        return t_dim
    return c_dim


def _tile_t(h, t, p, scope):
    n = scope + 'fc_t/fully_connected'
    if n + '/V' in p:
        t = fc_wn(t, p[n + '/V'], p[n + '/g'], p[n + '/biases'], 'lrelu')
    tt = t[:, None, None, :].expand(-1, h.shape[1], h.shape[2], -1)
    return torch.cat([h, tt], 3)


def init_sr_discriminator(ae_shape, latent_size, rng, dim=32, c_dim=None, t_dim=None):
    """c_dim: width of the cgan condition t (None: unconditional).  It enters at stage i == 3 only
    (cfl/models/blocks.py:182), i.e. for images of 64 pixels and more."""
    nb, start = _up_count(ae_shape)
    p = {}
    _conv_p(p, 'conv/Conv', rng, 4, 4, ae_shape[2], dim)
    for i in range(nb):
        s = 'conv%d/' % (i + 1)
        for j in range(4):
            _conv_p(p, s + _cname(j), rng, 3, 3, dim, dim)
        extra = _cond(p, s, rng, c_dim, t_dim) if (c_dim and i == 3) else 0
        _conv_p(p, s + _cname(4), rng, 4, 4, dim + extra, dim * 2)
        dim *= 2
    side_h, side_w = ae_shape[0], ae_shape[1]
    for _ in range(nb + 1):                      # every stride-2 'SAME' conv: ceil(n / 2)
        side_h, side_w = -(-side_h // 2), -(-side_w // 2)
    feat = side_h * side_w * dim
    _fc_p(p, 'disc_outputs/fully_connected', rng, feat, 1)
    _fc_p(p, 'latent_outputs/fully_connected', rng, feat, latent_size)
    return p


def sr_discriminator(x_flat, p, ae_shape, t=None):
    nb, _ = _up_count(ae_shape)
    h = x_flat.reshape((-1,) + tuple(ae_shape))
    h = CO.conv2d_weight_norm(h, p['conv/Conv/V'], p['conv/Conv/g'], p['conv/Conv/biases'], 2, 'lrelu')
    for i in range(nb):
        s = 'conv%d/' % (i + 1)
        for j in range(2):
            a, b = s + _cname(2 * j), s + _cname(2 * j + 1)
            r = CO.conv2d_weight_norm(h, p[a + '/V'], p[a + '/g'], p[a + '/biases'], 1, 'lrelu')
            r = CO.conv2d_weight_norm(r, p[b + '/V'], p[b + '/g'], p[b + '/biases'], 1, None)
            h = CO.lrelu(r + h)
        if t is not None and i == 3:
            h = _tile_t(h, t, p, s)
        c = s + _cname(4)
        h = CO.conv2d_weight_norm(h, p[c + '/V'], p[c + '/g'], p[c + '/biases'], 2, 'lrelu')
    f = h.reshape(h.shape[0], -1)
    d = 'disc_outputs/fully_connected'
    l = 'latent_outputs/fully_connected'
    return (fc_wn(f, p[d + '/V'], p[d + '/g'], p[d + '/biases']),
            fc_wn(f, p[l + '/V'], p[l + '/g'], p[l + '/biases']))


def init_convt_generator(ae_shape, in_dim, rng, dim=64, c_dim=None, t_dim=None):
    nb, start = _up_count(ae_shape)
    scale = 2 ** (nb - 1)
    p = {}
    in_dim = _gen_in(p, rng, in_dim, c_dim, t_dim)
    _fc_p(p, 'fc1/fully_connected', rng, in_dim, dim * scale * start * start)
    ci = dim * scale
    for i in range(nb - 1):
        co = dim * (2 ** (nb - i - 1))
        n = 'conv_t%d/Conv2d_transpose' % (i + 1)
        p[n + '/V'] = CO.xavier(rng, (5, 5, co, ci), 25 * co, 25 * ci)
        p[n + '/g'] = np.ones(co)
        p[n + '/biases'] = np.zeros(co)
        ci = co
    n = 'outputs/Conv2d_transpose'
    p[n + '/V'] = CO.xavier(rng, (5, 5, ae_shape[2], ci), 25 * ae_shape[2], 25 * ci)
    p[n + '/g'] = np.ones(ae_shape[2])
    p[n + '/biases'] = np.zeros(ae_shape[2])
    return p


def convt_generator(zc, p, ae_shape, data_type, dim=64, c_dim=None):
    nb, start = _up_count(ae_shape)
    scale = 2 ** (nb - 1)
    zc = _gen_cat(zc, p, c_dim)
    h = fc_wn(zc, p['fc1/fully_connected/V'], p['fc1/fully_connected/g'], p['fc1/fully_connected/biases'], 'relu')
    h = h.reshape(-1, start, start, dim * scale)
    for i in range(nb - 1):
        n = 'conv_t%d/Conv2d_transpose' % (i + 1)
        h = conv2d_transpose_weight_norm(h, p[n + '/V'], p[n + '/g'], p[n + '/biases'], 2, 'relu')
    n = 'outputs/Conv2d_transpose'
    h = conv2d_transpose_weight_norm(h, p[n + '/V'], p[n + '/g'], p[n + '/biases'], 2, None)
    return _data_act(h.reshape(h.shape[0], -1), data_type)


def init_conv_discriminator(ae_shape, latent_size, rng, dim=64, max_dim=512, c_dim=None, t_dim=None):
    """the cgan condition is concatenated after conv(nb-1) (cfl/models/blocks.py:382-395)"""
    nb, _ = _up_count(ae_shape)
    p = {}
    ci = ae_shape[2]
    h, w = ae_shape[0], ae_shape[1]
    for i in range(nb):
        _conv_p(p, 'conv%d/Conv' % (i + 1), rng, 5, 5, ci, dim)
        ci = dim
        if c_dim and i == nb - 2:
            ci += _cond(p, 'conv%d/' % (i + 1), rng, c_dim, t_dim)
        dim = min(dim * 2, max_dim)
        h, w = -(-h // 2), -(-w // 2)
    feat = h * w * ci
    _fc_p(p, 'disc_outputs/fully_connected', rng, feat, 1)
    _fc_p(p, 'latent_outputs/fully_connected', rng, feat, latent_size)
    return p


def conv_discriminator(x_flat, p, ae_shape, t=None):
    nb, _ = _up_count(ae_shape)
    h = x_flat.reshape((-1,) + tuple(ae_shape))
    for i in range(nb):
        n = 'conv%d/Conv' % (i + 1)
        h = CO.conv2d_weight_norm(h, p[n + '/V'], p[n + '/g'], p[n + '/biases'], 2, 'lrelu')
        if t is not None and i == nb - 2:
            h = _tile_t(h, t, p, 'conv%d/' % (i + 1))
    f = h.reshape(h.shape[0], -1)
    d = 'disc_outputs/fully_connected'
    l = 'latent_outputs/fully_connected'
    return (fc_wn(f, p[d + '/V'], p[d + '/g'], p[d + '/biases']),
            fc_wn(f, p[l + '/V'], p[l + '/g'], p[l + '/biases']))


GENERATORS = {'srgan': (init_sr_generator, sr_generator), 'conv': (init_convt_generator, convt_generator)}
DISCRIMINATORS = {'srgan': (init_sr_discriminator, sr_discriminator),
                  'conv': (init_conv_discriminator, conv_discriminator)}


# ---- losses (non-cgan branch) ----------------------------------------------------------------
def bce(logits, label):
    """mean sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))."""
    return (torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-logits.abs()))).mean()


def gan_losses(gp, dp, gan_type, ae_shape, data_type, real, enc_act, prj_c, neg_c, neg_tgt_act, z, eps,
               lambda_gp, lambda_dra, m_enc, m_prj):
    """Returns (d_total, g_total, parts).  All encoder-side inputs are constants here: both
    optimisers only touch generator / discriminator variables (cfl/models/cfl.py:1087-1096).
      real        data_unlabeled_ae_target [B, prod(ae_shape)]
      enc_act     s_encoder.activations       (dst encoder of the unlabeled target latent)
      prj_c       s_g.one_prototype_activations
      neg_c       s_neg_src.one_prototype_activations
      neg_tgt_act s_neg_target.activations"""
    gen = GENERATORS[gan_type][1]
    disc = DISCRIMINATORS[gan_type][1]
    g = gen(torch.cat([z, enc_act], 1), gp, ae_shape, data_type)
    g_neg = gen(torch.cat([z, neg_c], 1), gp, ae_shape, data_type)
    g_prj = gen(torch.cat([z, prj_c], 1), gp, ae_shape, data_type)
    d_real, l_real = disc(real, dp, ae_shape)
    d_fake, l_fake = disc(g, dp, ae_shape)
    d_prj, _ = disc(g_prj, dp, ae_shape)
    _, l_neg = disc(g_neg, dp, ae_shape)
    parts = {}
    one, zero = 1.0, 0.0
    parts['d_loss_real'] = bce(d_real, one)
    parts['d_loss_fake'] = 0.5 * (bce(d_fake, zero) + bce(d_prj, zero))
    d_total = parts['d_loss_real'] + parts['d_loss_fake']
    if lambda_gp:
        std = torch.sqrt(real.var(unbiased=False))
        x_hat = (real + lambda_dra * std * eps).detach().requires_grad_(True)
        d_hat, _ = disc(x_hat, dp, ae_shape)
        grad, = torch.autograd.grad(d_hat.sum(), x_hat, create_graph=True)
        parts['d_grad_loss'] = lambda_gp * ((torch.sqrt((grad * grad).sum(1)) - 1.0) ** 2).mean()
        d_total = d_total + parts['d_grad_loss']
    parts['d_loss_d'] = ((l_real - enc_act) ** 2).sum(-1).mean()
    d_total = d_total + parts['d_loss_d']

    parts['g_loss'] = 0.5 * (bce(d_prj, one) + bce(d_fake, one))
    g_total = parts['g_loss']
    g_d_enc = ((l_fake - enc_act) ** 2).sum(-1)
    g_d_neg = ((l_neg - neg_tgt_act) ** 2).sum(-1)
    if m_enc:
        parts['g_loss_d'] = (torch.clamp(torch.sqrt(g_d_enc + 1e-7) - m_enc, min=0) ** 2).mean()
    else:
        parts['g_loss_d'] = g_d_enc.mean()
    g_total = g_total + parts['g_loss_d']
    if m_prj:
        parts['g_loss_d_neg'] = (torch.clamp(m_prj - torch.sqrt(g_d_neg + 1e-7), min=0) ** 2).mean()
        g_total = g_total + parts['g_loss_d_neg']
    parts['d_real_accuracy'] = (d_real > 0).double().mean()
    parts['d_fake_accuracy'] = (d_fake <= 0).double().mean()
    parts['g_accuracy'] = (d_fake > 0).double().mean()
    parts['g_activations'] = g
    return d_total, g_total, parts


def cgan_losses(gp, dp, gan_type, ae_shape, data_type, real_pos, real_neg, pos_c, neg_c, z, eps, lambda_gp,
                lambda_dra):
    """The --cgan branch (cfl/models/cfl.py:747-782, 969-981, 1022-1038).
      real_pos / real_neg  data_pos_ae_target / data_neg_ae_target [B, prod(ae_shape)]
      pos_c / neg_c        the condition: source encoder activations, or the raw source latents with --t-dim"""
    gen = GENERATORS[gan_type][1]
    disc = DISCRIMINATORS[gan_type][1]
    B, cd = pos_c.shape
    g = gen(torch.cat([z, pos_c], 1), gp, ae_shape, data_type, c_dim=cd)
    half = (pos_c[:B // 2] + pos_c[B // 2:]) / 2.0
    g_int = gen(torch.cat([z[:B // 2], half], 1), gp, ae_shape, data_type, c_dim=cd)
    d_real, _ = disc(real_pos, dp, ae_shape, t=pos_c)
    d_fake, _ = disc(g, dp, ae_shape, t=pos_c)
    d_neg, _ = disc(real_neg, dp, ae_shape, t=neg_c)
    d_int, _ = disc(g_int, dp, ae_shape, t=half)
    parts = {}
    parts['d_loss_real'] = bce(d_real, 1.0)
    parts['d_loss_fake'] = bce(d_fake, 0.0)
    parts['d_loss_neg'] = bce(d_neg, 0.0)
    d_total = parts['d_loss_real'] + (parts['d_loss_fake'] + parts['d_loss_neg']) / 2.0
    if lambda_gp:
        std = torch.sqrt(real_pos.var(unbiased=False))
        x_hat = (real_pos + lambda_dra * std * eps).detach().requires_grad_(True)
        d_hat, _ = disc(x_hat, dp, ae_shape, t=pos_c)
        grad, = torch.autograd.grad(d_hat.sum(), x_hat, create_graph=True)
        parts['d_grad_loss'] = lambda_gp * ((torch.sqrt((grad * grad).sum(1)) - 1.0) ** 2).mean()
        d_total = d_total + parts['d_grad_loss']
    parts['g_loss'] = bce(d_fake, 1.0)
    parts['g_loss_int'] = bce(d_int, 1.0)
    g_total = parts['g_loss'] + parts['g_loss_int']
    parts['d_real_accuracy'] = (d_real > 0).double().mean()
    parts['d_fake_accuracy'] = (d_fake <= 0).double().mean()
    parts['g_accuracy'] = (d_fake > 0).double().mean()
    return d_total, g_total, parts


class AdamTF:
    """TF-1 AdamOptimizer over a dict of tensors (cfl/models/cfl.py:1090-1096)."""

    def __init__(self, params, lr, beta1, beta2=0.999, eps=1e-8):
        # the hyper-parameters are float32 tensors in the reference's graph (see oracle/cfl_oracle.py AdamState.hyper)
        self.lr, self.b1, self.b2, self.eps = (float(np.float32(x)) for x in (lr, beta1, beta2, eps))
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.b1p, self.b2p = np.float32(beta1), np.float32(beta2)

    def lr_t(self):
        # the power accumulators are float32 variables in TF; the expression is evaluated in the dtype of the
        # trained variables (float64 here), like oracle/cfl_oracle.py AdamState.lr_t
        return float(np.float64(self.lr) * np.sqrt(1.0 - np.float64(self.b2p)) / (1.0 - np.float64(self.b1p)))

    def apply(self, params, grads):
        lr_t = self.lr_t()
        for k in params:
            g = grads[k]
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            params[k] = params[k] - lr_t * self.m[k] / (torch.sqrt(self.v[k]) + self.eps)
        self.b1p = np.float32(self.b1p * np.float32(self.b1))
        self.b2p = np.float32(self.b2p * np.float32(self.b2))


class GanOracle:
    def __init__(self, gan_type, ae_shape, data_type, z_dim, latent_size, seed=0, d_lr=2e-4, d_beta1=0.5,
                 d_beta2=0.999, g_lr=2e-4, g_beta1=0.5, g_beta2=0.999, lambda_gp=0.5, lambda_dra=0.5,
                 m_enc=None, m_prj=None, dtype=torch.float64, cgan=False, c_dim=None, t_dim=None):
        rng = np.random.RandomState(seed)
        self.gan_type, self.ae_shape, self.data_type = gan_type, tuple(ae_shape), data_type
        self.cgan = cgan
        if cgan:
            c_dim = c_dim or latent_size
            self.cfg = dict(lambda_gp=lambda_gp, lambda_dra=lambda_dra)
            gp = GENERATORS[gan_type][0](self.ae_shape, z_dim + c_dim, rng, c_dim=c_dim, t_dim=t_dim)
            dp = DISCRIMINATORS[gan_type][0](self.ae_shape, latent_size, rng, c_dim=c_dim, t_dim=t_dim)
        else:
            self.cfg = dict(lambda_gp=lambda_gp, lambda_dra=lambda_dra, m_enc=m_enc, m_prj=m_prj)
            gp = GENERATORS[gan_type][0](self.ae_shape, z_dim + latent_size, rng)
            dp = DISCRIMINATORS[gan_type][0](self.ae_shape, latent_size, rng)
        self.gp = {k: torch.tensor(v, dtype=dtype) for k, v in gp.items()}
        self.dp = {k: torch.tensor(v, dtype=dtype) for k, v in dp.items()}
        self.g_adam = AdamTF(self.gp, g_lr, g_beta1, g_beta2)
        self.d_adam = AdamTF(self.dp, d_lr, d_beta1, d_beta2)

    def losses_and_grads(self, *batch):
        """non-cgan: (real, enc_act, prj_c, neg_c, neg_tgt_act, z, eps); cgan: (real_pos, real_neg, pos_c, neg_c, z, eps)"""
        gp = {k: v.detach().requires_grad_(True) for k, v in self.gp.items()}
        dp = {k: v.detach().requires_grad_(True) for k, v in self.dp.items()}
        fn = cgan_losses if self.cgan else gan_losses
        d_total, g_total, parts = fn(gp, dp, self.gan_type, self.ae_shape, self.data_type, *batch, **self.cfg)
        dk, gk = list(dp), list(gp)
        dg = torch.autograd.grad(d_total, [dp[k] for k in dk], retain_graph=True, allow_unused=True)
        gg = torch.autograd.grad(g_total, [gp[k] for k in gk], allow_unused=True)
        d_grads = {k: (t if t is not None else torch.zeros_like(dp[k])) for k, t in zip(dk, dg)}
        g_grads = {k: (t if t is not None else torch.zeros_like(gp[k])) for k, t in zip(gk, gg)}
        parts = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in parts.items()}
        return d_total.detach(), g_total.detach(), parts, d_grads, g_grads

    def step(self, *batch):
        d_total, g_total, parts, d_grads, g_grads = self.losses_and_grads(*batch)
        self.d_adam.apply(self.dp, d_grads)
        self.g_adam.apply(self.gp, g_grads)
        return d_total, g_total, parts
