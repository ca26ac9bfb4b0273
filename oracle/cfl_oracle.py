"""CPU oracle for the cfl pair-distance training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.

PARITY: COMPOSITION PINNED, TENSORFLOW KERNELS UNPINNED ("parity unpinned" for
the TF primitives).  The reference (appier/compatibility-family-learning) runs
this arithmetic inside TensorFlow 1.x, which is not installable in the build
container, and ships no tests or golden vectors for it.
What IS pinned: tests/golden/make_arith_goldens.py imports the reference's own
``cfl.models.dist.Dist`` / ``cfl.models.cfl.CFL`` classes (with ``cfl.layers``,
``cfl.ops``, ``cfl.models.base``, ``cfl.models.blocks`` behind them) and runs
their constructors over an eager float64 stand-in for the TF primitives
(tests/golden/tf_standin.py); tests/test_arith_goldens.py requires this file to
reproduce every distance, score, loss part, gradient (by TF variable name) and
every variable after 1-3 Adam steps of 11 linear cases to float64 round-off.
So which head feeds which side, reshape orders, softmax axes, loss / pos_weight
/ regulariser placement, optimiser ownership and the maximum / relu tie rules
follow the reference's lines; only the primitives themselves (matmul, softmax,
sigmoid_cross_entropy_with_logits, Adam, ... restated from TF's documentation)
were never compared with a real TensorFlow run.  Further pins: the analytic
known-answer tests of SURVEY.md App. A.7, float64 finite differences of every
analytic gradient and a torch-autograd cross-check (tests/test_oracle.py).  The
data / eval side (the host restatements in compatibility-family-learning_amd/cfl/input_data.py and
cfl/bin/evaluate_total.py; dist_eval below) is pinned against vectors captured by importing the reference
unmodified (tests/golden/make_data_goldens.py, tests/test_input_data.py, tests/test_evaluate_total.py).

All citations are ``path:line`` relative to the reference tree.

Conventions
-----------
* A training *row* is two independent pairs: (pos_src, pos_dst, neg_src,
  neg_dst), each ``[B, D]`` float (cfl/input_data.py:542-589).
* ``src`` is the encoder whose ``build_dist`` is called, ``dst`` is its
  ``target`` argument (cfl/models/base.py:107, cfl/models/dist.py:70).
* Weights are kept in the reference (TensorFlow) layout ``[D, N]``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np

THRESHOLD_FLOOR = 1e-6  # cfl/models/blocks.py:19-21


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
@dataclass
class EncoderCfg:
    """Shape/flag set of one distance encoder.

    ``style='dist'``  : FCEncoder of cfl/models/dist.py:12-68 (plain FC, biases on
                        both heads, L2 reg on weights *and* biases).
    ``style='cfl'``   : FCPCD + DistBase.build_prototypes, cfl/models/blocks.py:
                        477-527 + cfl/models/base.py:43-105 (weight-normalised
                        FC; biases only when dist_type starts with 'pcd').
    """
    D: int
    L: int
    K: int
    dist_type: str = 'pcd'          # 'pcd' | 'monomer' | 'siamese'
    style: str = 'dist'             # 'dist' | 'cfl'
    act_type: Optional[str] = None  # None/'linear' | 'sigmoid' | 'tanh' | 'relu'

    @property
    def weight_norm(self) -> bool:
        return self.style == 'cfl'

    @property
    def has_bias(self) -> bool:
        # cfl/models/base.py:45-46,62-63 ; cfl/models/dist.py:52,65
        return self.style == 'dist' or self.dist_type.startswith('pcd')

    @property
    def has_proto(self) -> bool:
        return self.dist_type in ('pcd', 'monomer')


@dataclass
class LossCfg:
    """cfl/models/cfl.py:868-949 ; cfl/models/dist.py:253-284."""
    use_threshold: bool = True
    pos_weight: Optional[float] = None
    caffe_margin: Optional[float] = None
    lambda_m: float = 0.0
    reg_const: float = 0.0


# --------------------------------------------------------------------------
# parameter construction
# --------------------------------------------------------------------------
def xavier_uniform(rng: np.random.RandomState, fan_in: int, fan_out: int,
                   dtype=np.float32) -> np.ndarray:
    """tf.contrib.layers.xavier_initializer(): U(+-sqrt(6/(fan_in+fan_out)))."""
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


def init_encoder_params(cfg: EncoderCfg, rng: np.random.RandomState,
                        dtype=np.float32) -> Dict[str, np.ndarray]:
    """Variables of one encoder, named after SURVEY.md App. D (scope-relative).

    'outputs/W' is ``latent_outputs`` for style 'dist' (cfl/models/dist.py:43-53)
    and ``outputs`` for style 'cfl' (cfl/models/base.py:44-57).
    """
    p: Dict[str, np.ndarray] = {}
    p['outputs/W'] = xavier_uniform(rng, cfg.D, cfg.L, dtype)
    if cfg.weight_norm:
        p['outputs/g'] = np.ones(cfg.L, dtype)
    if cfg.has_bias:
        p['outputs/b'] = np.zeros(cfg.L, dtype)
    if cfg.has_proto:
        n = cfg.L * cfg.K
        p['proto/W'] = xavier_uniform(rng, cfg.D, n, dtype)
        if cfg.weight_norm:
            p['proto/g'] = np.ones(n, dtype)
        if cfg.has_bias:
            p['proto/b'] = np.zeros(n, dtype)
    if cfg.dist_type == 'monomer':
        # cfl/models/base.py:94-105: input = `outputs` (L), N = K, no bias
        p['mono/W'] = xavier_uniform(rng, cfg.L, cfg.K, dtype)
        if cfg.weight_norm:
            p['mono/g'] = np.ones(cfg.K, dtype)
    return p


# --------------------------------------------------------------------------
# element-wise pieces
# --------------------------------------------------------------------------
def normalize(x, scale, shift=0.0, clip_min=None, clip_max=None):
    """cfl/ops.py:198-202 : ``x / scale + shift`` then optional clip."""
    y = x / x.dtype.type(scale) + x.dtype.type(shift)
    if clip_min is not None or clip_max is not None:
        y = np.clip(y, clip_min, clip_max)
    return y


def normalize_v2(x, scale=None, mean=None, norm=None, clip_min=None,
                 clip_max=None):
    """cfl/ops.py:66-124, scalar mean/norm branch (vector inputs)."""
    t = x.dtype.type
    y = x
    if scale is not None and scale != 1.0:
        y = y * t(scale)
    if mean is not None and mean != 0.0:
        y = y - t(mean)
    if norm is not None and norm != 1.0:
        y = y / t(norm)
    if clip_min is not None and clip_max is None:
        y = np.maximum(t(clip_min), y)
    elif clip_min is None and clip_max is not None:
        y = np.minimum(t(clip_max), y)
    elif clip_min is not None and clip_max is not None:
        y = np.clip(y, t(clip_min), t(clip_max))
    return y


def _act(y, act_type):
    if act_type in (None, 'linear'):
        return y
    if act_type == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-y))
    if act_type == 'tanh':
        return np.tanh(y)
    if act_type == 'relu':
        return np.maximum(y, 0)
    raise ValueError(act_type)


def _act_grad(y, a, act_type):
    """d act / d y given pre-activation y and activation a."""
    if act_type in (None, 'linear'):
        return np.ones_like(y)
    if act_type == 'sigmoid':
        return a * (1 - a)
    if act_type == 'tanh':
        return 1 - a * a
    if act_type == 'relu':
        return (y > 0).astype(y.dtype)
    raise ValueError(act_type)


def softmax(z, axis=-1):
    z = z - z.max(axis=axis, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=axis, keepdims=True)


def bce_with_logits(x, z):
    """tf.nn.sigmoid_cross_entropy_with_logits stable form."""
    return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))


def sigmoid(x):
    return 0.5 * (1.0 + np.tanh(0.5 * x))


# --------------------------------------------------------------------------
# heads
# --------------------------------------------------------------------------
def fc_head(x, p, name, weight_norm):
    """Linear head.

    plain : cfl/models/dist.py:45-65       y = x W + b
    wn    : cfl/layers.py:80-90            y = (x V) * (g / sqrt(sum_rows V^2)) + b
    Returns (y, cache) with cache = (xV, scaler) for the backward.
    """
    W = p[name + '/W']
    xv = x @ W
    if weight_norm:
        n = np.sqrt((W * W).sum(axis=0))
        s = p[name + '/g'] / n
        y = xv * s[None, :]
    else:
        s = None
        y = xv
    b = p.get(name + '/b')
    if b is not None:
        y = y + b[None, :]
    return y, (xv, s)


def fc_head_bwd(x, p, name, weight_norm, cache, dy, grads):
    """Accumulates d/dW (or V), d/dg, d/db into ``grads``; returns dx.

    Weight-norm backward: SURVEY App. A.5 last paragraph.
    """
    W = p[name + '/W']
    xv, s = cache
    if name + '/b' in p:
        grads[name + '/b'] = grads.get(name + '/b', 0) + dy.sum(axis=0)
    if weight_norm:
        g = p[name + '/g']
        n = np.sqrt((W * W).sum(axis=0))
        c = (dy * xv).sum(axis=0)
        grads[name + '/g'] = grads.get(name + '/g', 0) + c / n
        dxv = dy * s[None, :]
        dW = x.T @ dxv - (g * c / n ** 3)[None, :] * W
    else:
        dxv = dy
        dW = x.T @ dy
    grads[name + '/W'] = grads.get(name + '/W', 0) + dW
    return dxv @ W.T


# --------------------------------------------------------------------------
# distances (cfl/models/base.py:107-146 == cfl/models/dist.py:70-89)
# --------------------------------------------------------------------------
def dist_pcd(v, P):
    """v: [B,L] (dst activations), P: [B,K,L] (src prototypes) -> d [B]."""
    K = P.shape[1]
    if K > 1:
        diff = v[:, None, :] - P
        logits = -(diff * diff).sum(-1)
        s = softmax(logits, -1)
        m = (P * s[:, :, None]).sum(-2)
        r = v - m
        return (r * r).sum(-1), (s, m)
    diff = v - P[:, 0, :]
    return (diff * diff).sum(-1), None


def dist_pcd_bwd(v, P, cache, dd):
    """SURVEY App. A.5; dd = dL/dd [B]. Returns (dv [B,L], dP [B,K,L])."""
    K = P.shape[1]
    if K > 1:
        s, m = cache
        r = v - m
        q = -2.0 * (r[:, None, :] * P).sum(-1)                 # [B,K]
        qbar = (s * q).sum(-1, keepdims=True)
        dl = s * (q - qbar)                                     # dd/dlogit_k
        vmP = v[:, None, :] - P                                 # [B,K,L]
        dv = 2.0 * r + (dl[:, :, None] * (-2.0) * vmP).sum(1)
        dP = -2.0 * s[:, :, None] * r[:, None, :] + dl[:, :, None] * 2.0 * vmP
    else:
        diff = v - P[:, 0, :]
        dv = 2.0 * diff
        dP = (-2.0 * diff)[:, None, :]
    return dv * dd[:, None], dP * dd[:, None, None]


def dist_monomer(a, u, P):
    """a: [B,L] src activations, u: [B,K] src monomer_outputs (pre-softmax),
    P: [B,K,L] dst prototypes.  cfl/models/base.py:109-117."""
    w = softmax(u, -1)
    diff = a[:, None, :] - P
    e = (diff * diff).sum(-1)
    return (w * e).sum(-1), (w, e)


def dist_monomer_bwd(a, u, P, cache, dd):
    w, e = cache
    d = (w * e).sum(-1, keepdims=True)
    amP = a[:, None, :] - P
    da = (w[:, :, None] * 2.0 * amP).sum(1)
    dP = -2.0 * w[:, :, None] * amP
    du = w * (e - d)
    return da * dd[:, None], du * dd[:, None], dP * dd[:, None, None]


def dist_siamese(a, b):
    diff = a - b
    return (diff * diff).sum(-1)


# --------------------------------------------------------------------------
# one side-pair forward/backward
# --------------------------------------------------------------------------
def _pair_forward(cfg: EncoderCfg, p_src, p_dst, xs, xt):
    """Distance of a batch of (src, dst) pairs. Mirrors TF's pruning of unused
    heads: pcd uses src.proto + dst.outputs; monomer uses
    src.outputs, src.mono, dst.proto; siamese uses outputs on both sides."""
    wn = cfg.weight_norm
    B = xs.shape[0]
    c = {}
    if cfg.dist_type.startswith('pcd'):
        yP, c['P'] = fc_head(xs, p_src, 'proto', wn)
        yv, c['v'] = fc_head(xt, p_dst, 'outputs', wn)
        aP = _act(yP, cfg.act_type)
        av = _act(yv, cfg.act_type)
        d, c['d'] = dist_pcd(av, aP.reshape(B, cfg.K, cfg.L))
        c.update(yP=yP, yv=yv, aP=aP, av=av)
    elif cfg.dist_type == 'monomer':
        ya, c['a'] = fc_head(xs, p_src, 'outputs', wn)
        yu, c['u'] = fc_head(ya, p_src, 'mono', wn)   # pre-activation `outputs`
        yP, c['P'] = fc_head(xt, p_dst, 'proto', wn)
        aa = _act(ya, cfg.act_type)
        aP = _act(yP, cfg.act_type)
        d, c['d'] = dist_monomer(aa, yu, aP.reshape(B, cfg.K, cfg.L))
        c.update(ya=ya, yu=yu, yP=yP, aa=aa, aP=aP)
    elif cfg.dist_type == 'siamese':
        ya, c['a'] = fc_head(xs, p_src, 'outputs', wn)
        yb, c['b'] = fc_head(xt, p_dst, 'outputs', wn)
        aa = _act(ya, cfg.act_type)
        ab = _act(yb, cfg.act_type)
        d = dist_siamese(aa, ab)
        c.update(ya=ya, yb=yb, aa=aa, ab=ab)
    else:
        raise ValueError(cfg.dist_type)
    return d, c


def _pair_backward(cfg: EncoderCfg, p_src, p_dst, xs, xt, c, dd, g_src, g_dst):
    wn = cfg.weight_norm
    B = xs.shape[0]
    if cfg.dist_type.startswith('pcd'):
        dv, dP = dist_pcd_bwd(c['av'], c['aP'].reshape(B, cfg.K, cfg.L),
                              c['d'], dd)
        dP = dP.reshape(B, cfg.K * cfg.L)
        dyP = dP * _act_grad(c['yP'], c['aP'], cfg.act_type)
        dyv = dv * _act_grad(c['yv'], c['av'], cfg.act_type)
        fc_head_bwd(xs, p_src, 'proto', wn, c['P'], dyP, g_src)
        fc_head_bwd(xt, p_dst, 'outputs', wn, c['v'], dyv, g_dst)
    elif cfg.dist_type == 'monomer':
        da, du, dP = dist_monomer_bwd(c['aa'], c['yu'],
                                      c['aP'].reshape(B, cfg.K, cfg.L),
                                      c['d'], dd)
        dP = dP.reshape(B, cfg.K * cfg.L)
        dya = da * _act_grad(c['ya'], c['aa'], cfg.act_type)
        dya = dya + fc_head_bwd(c['ya'], p_src, 'mono', wn, c['u'], du, g_src)
        dyP = dP * _act_grad(c['yP'], c['aP'], cfg.act_type)
        fc_head_bwd(xs, p_src, 'outputs', wn, c['a'], dya, g_src)
        fc_head_bwd(xt, p_dst, 'proto', wn, c['P'], dyP, g_dst)
    else:
        diff = c['aa'] - c['ab']
        da = 2.0 * diff * dd[:, None]
        db = -da
        dya = da * _act_grad(c['ya'], c['aa'], cfg.act_type)
        dyb = db * _act_grad(c['yb'], c['ab'], cfg.act_type)
        fc_head_bwd(xs, p_src, 'outputs', wn, c['a'], dya, g_src)
        fc_head_bwd(xt, p_dst, 'outputs', wn, c['b'], dyb, g_dst)


def pair_scores(cfg: EncoderCfg, params, raw_thr, xs, xt, params_dst=None):
    """score = max(thr,1e-6) - d, shape [n] (cfl/models/blocks.py:18-22;
    the value cfl/utils.py:245 fetches as ``val_s_pos_predicts.outputs``)."""
    p_dst = params if params_dst is None else params_dst
    d, _ = _pair_forward(cfg, params, p_dst, xs, xt)
    thr = np.maximum(raw_thr, d.dtype.type(THRESHOLD_FLOOR))
    return thr - d


# --------------------------------------------------------------------------
# the training step
# --------------------------------------------------------------------------
def reg_loss_and_grad(cfg: EncoderCfg, p, reg_const, grads=None):
    """tf.contrib.layers.l2_regularizer(s)(w) = s * sum(w^2) / 2 over every
    regularised variable: V/W and biases of the heads that exist, V only for
    monomer_outputs, never g (cfl/models/base.py:51-53,69-72,
    101; cfl/models/dist.py:49-50,62-64).  NOTE: TF creates the regulariser
    term when the variable is created, so *every* head variable of the encoder
    contributes, whether or not the head is used by the distance."""
    tot = 0.0
    if not reg_const:
        return tot
    for k, w in p.items():
        if k.endswith('/g'):
            continue
        tot = tot + reg_const * 0.5 * (w * w).sum()
        if grads is not None:
            grads[k] = grads.get(k, 0) + reg_const * w
    return tot


def train_step_loss_and_grads(cfg: EncoderCfg, lcfg: LossCfg, params, raw_thr,
                              batch, params_dst=None):
    """Forward + analytic backward of one training row-batch.

    batch = (pos_src, pos_dst, neg_src, neg_dst), already normalised.
    Returns (scalars dict, grads dict, grads_dst dict|None, dthr, dthr_aux).

    ``dthr``     : gradient of s_total_loss w.r.t. the raw threshold variable
                   (non-zero only when the threshold loss is part of the total,
                   cfl/models/cfl.py:890-893,1080-1085).
    ``dthr_aux`` : gradient of s_thres_loss w.r.t. the raw threshold, used by the
                   separate ``th_optim`` when use_threshold is off
                   (cfl/models/cfl.py:1076-1079).
    """
    xps, xpd, xns, xnd = batch
    dt = xps.dtype.type
    directed = params_dst is not None
    p_dst = params_dst if directed else params
    B = xps.shape[0]

    d_pos, c_pos = _pair_forward(cfg, params, p_dst, xps, xpd)
    d_neg, c_neg = _pair_forward(cfg, params, p_dst, xns, xnd)

    thr = np.maximum(raw_thr, dt(THRESHOLD_FLOOR))
    thr_mask = dt(1.0) if raw_thr >= dt(THRESHOLD_FLOOR) else dt(0.0)
    o_pos = thr - d_pos
    o_neg = thr - d_neg
    l_pos = bce_with_logits(o_pos, dt(1.0)).mean()
    l_neg = bce_with_logits(o_neg, dt(0.0)).mean()
    pw = dt(lcfg.pos_weight) if lcfg.pos_weight else dt(1.0)
    l_thr = l_pos * pw + l_neg                       # cfl.py:887-891

    grads: Dict[str, np.ndarray] = {}
    grads_dst: Optional[Dict[str, np.ndarray]] = {} if directed else None
    g_dst = grads_dst if directed else grads

    l_reg = reg_loss_and_grad(cfg, params, lcfg.reg_const, grads)
    if directed:
        l_reg = l_reg + reg_loss_and_grad(cfg, params_dst, lcfg.reg_const,
                                          grads_dst)
    total = l_reg

    # dL/do for the BCE terms
    do_pos = (sigmoid(o_pos) - 1.0) * pw / B
    do_neg = sigmoid(o_neg) / B
    dthr_aux = (do_pos.sum() + do_neg.sum()) * thr_mask

    dd_pos = np.zeros_like(d_pos)
    dd_neg = np.zeros_like(d_neg)
    dthr = dt(0.0)
    if lcfg.use_threshold:
        total = total + l_thr
        dd_pos += -do_pos
        dd_neg += -do_neg
        dthr = dthr_aux

    l_cd = dt(0.0)
    if lcfg.caffe_margin:
        m = dt(lcfg.caffe_margin)
        cd_pos = d_pos.mean() * pw
        cd_neg = np.maximum(dt(0.0), m - d_neg).mean()
        l_cd = 0.5 * (cd_pos + cd_neg)               # cfl.py:912-921
        dd_pos += 0.5 * pw / B
        dd_neg += -0.5 * (d_neg < m).astype(d_neg.dtype) / B
        total = total + l_cd
    elif lcfg.lambda_m:
        l_cd = d_pos.mean() * dt(lcfg.lambda_m) * pw  # cfl.py:922-929
        dd_pos += pw * dt(lcfg.lambda_m) / B
        total = total + l_cd

    _pair_backward(cfg, params, p_dst, xps, xpd, c_pos, dd_pos, grads, g_dst)
    _pair_backward(cfg, params, p_dst, xns, xnd, c_neg, dd_neg, grads, g_dst)

    acc = 0.5 * ((o_pos > 0).mean() + (o_neg <= 0).mean())   # cfl.py:932-937
    scalars = dict(total=total, reg=l_reg, thres=l_thr, loss_pos=l_pos,
                   loss_neg=l_neg, cd=l_cd, accuracy=acc,
                   mean_d_pos=d_pos.mean(), mean_d_neg=d_neg.mean(),
                   mean_o_pos=o_pos.mean(), mean_o_neg=o_neg.mean(),
                   margins=(d_pos - d_neg).mean(),
                   d_pos=d_pos, d_neg=d_neg)
    return scalars, grads, grads_dst, dthr, dthr_aux


# --------------------------------------------------------------------------
# TF-1.x Adam
# --------------------------------------------------------------------------
@dataclass
class AdamState:
    """tf.train.AdamOptimizer slots: m, v per variable + the two power
    accumulators (kept in float32 like TF's ``beta1_power``/``beta2_power``
    variables)."""
    lr: float
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    m: Dict[str, np.ndarray] = field(default_factory=dict)
    v: Dict[str, np.ndarray] = field(default_factory=dict)
    beta1_power: np.float32 = None
    beta2_power: np.float32 = None

    def __post_init__(self):
        if self.beta1_power is None:
            self.beta1_power = np.float32(self.beta1)
            self.beta2_power = np.float32(self.beta2)

    def hyper(self, dtype):
        """(lr, beta1, beta2, eps) as the reference's graph holds them: python floats converted to float32 tensors
        (the variables are float32), THEN taken to the evaluation dtype.  beta2 = 0.999 is not a float32 number
        (fl32(0.999) = 0.99900001287...), and TensorFlow's Adam kernel forms (1 - beta2) from that tensor, so the
        v update weighs g^2 by 0.00099998713, not by 0.001: a 1.3e-5 relative difference that a float64
        evaluation of the same graph must carry too, or Adam's first (sign-like, lr-sized) steps of a
        high-dimensional batch separate the trajectories at the 1e-5 level."""
        t = np.dtype(dtype).type
        return tuple(t(np.float32(x)) for x in (self.lr, self.beta1, self.beta2, self.eps))

    def lr_t(self, dtype=np.float32):
        """lr * sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in ``dtype``."""
        t = np.dtype(dtype).type
        lr = self.hyper(dtype)[0]
        return lr * np.sqrt(t(1) - t(self.beta2_power)) / (
            t(1) - t(self.beta1_power))

    def apply(self, params: Dict[str, np.ndarray], grads: Dict[str, np.ndarray]):
        lr_t = None
        for k, g in grads.items():
            th = params[k]
            t = th.dtype.type
            if lr_t is None:
                lr_t = self.lr_t(th.dtype)
                _, b1, b2, eps = self.hyper(th.dtype)
            g = np.asarray(g, dtype=th.dtype)
            m = self.m.setdefault(k, np.zeros_like(th))
            v = self.v.setdefault(k, np.zeros_like(th))
            m[...] = b1 * m + (t(1) - b1) * g
            v[...] = b2 * v + (t(1) - b2) * g * g
            params[k] = th - lr_t * m / (np.sqrt(v) + eps)
        self.beta1_power = np.float32(self.beta1_power * np.float32(self.beta1))
        self.beta2_power = np.float32(self.beta2_power * np.float32(self.beta2))


def adam_tf_flat(theta, m, v, g, lr_t, beta1, beta2, eps):
    """Flat-array form used to check the HIP multi-tensor Adam kernel."""
    t = theta.dtype.type
    b1, b2, eps = (t(np.float32(x)) for x in (beta1, beta2, eps))     # float32 tensors in the reference's graph
    m_new = b1 * m + (t(1) - b1) * g
    v_new = b2 * v + (t(1) - b2) * g * g
    theta_new = theta - t(lr_t) * m_new / (np.sqrt(v_new) + eps)
    return theta_new, m_new, v_new


class OracleTrainer:
    """End-to-end CPU training loop of the Monomer-data model ``Dist``
    (cfl/models/dist.py:92-327 + cfl/bin/train_dist.py:77-87) and of the linear
    ``CFL`` dist phase (cfl/models/cfl.py:1399-1414), parameterised the same."""

    def __init__(self, cfg: EncoderCfg, lcfg: LossCfg, lr=1e-3, beta1=0.9,
                 beta2=0.999, seed=0, dtype=np.float32, directed=False,
                 params=None, params_dst=None):
        rng = np.random.RandomState(seed)
        self.cfg, self.lcfg, self.dtype = cfg, lcfg, dtype
        self.params = params if params is not None else init_encoder_params(
            cfg, rng, dtype)
        self.params_dst = params_dst
        if directed and params_dst is None:
            self.params_dst = init_encoder_params(cfg, rng, dtype)
        self.raw_thr = np.dtype(dtype).type(THRESHOLD_FLOOR)
        self.adam = AdamState(lr, beta1, beta2)
        self.adam_th = AdamState(lr, beta1, beta2)  # th_optim (cfl.py:1076)

    def _flat(self):
        flat = {'src/' + k: v for k, v in self.params.items()}
        if self.params_dst is not None:
            flat.update({'dst/' + k: v for k, v in self.params_dst.items()})
        return flat

    def step(self, batch):
        batch = tuple(np.asarray(b, dtype=self.dtype) for b in batch)
        sc, g, gd, dthr, dthr_aux = train_step_loss_and_grads(
            self.cfg, self.lcfg, self.params, self.raw_thr, batch,
            self.params_dst)
        flat = self._flat()
        grads = {'src/' + k: v for k, v in g.items()}
        if gd is not None:
            grads.update({'dst/' + k: v for k, v in gd.items()})
        # variables without any gradient path are not in TF's var_list update
        # only if their gradient is None; heads that exist but are unused get
        # None grads (skipped by Adam) unless regularised.
        flat['thr'] = np.asarray(self.raw_thr)
        if self.lcfg.use_threshold:
            grads['thr'] = np.asarray(dthr, dtype=self.dtype)
            self.adam.apply(flat, grads)
        else:
            self.adam.apply(flat, grads)
            th = {'thr': flat['thr']}
            self.adam_th.apply(th, {'thr': np.asarray(dthr_aux, self.dtype)})
            flat['thr'] = th['thr']
        self.raw_thr = np.dtype(self.dtype).type(flat['thr'])
        for k in self.params:
            self.params[k] = flat['src/' + k]
        if self.params_dst is not None:
            for k in self.params_dst:
                self.params_dst[k] = flat['dst/' + k]
        return sc

    def scores(self, xs, xt):
        return pair_scores(self.cfg, self.params, self.raw_thr,
                           np.asarray(xs, self.dtype), np.asarray(xt, self.dtype),
                           self.params_dst)


def dist_eval(score_pos, score_neg):
    """cfl/utils.py:227-274 given the two score vectors."""
    from sklearn.metrics import roc_auc_score
    total = score_pos.shape[0] + score_neg.shape[0]
    correct = int((score_pos > 0).sum()) + int((score_neg <= 0).sum())
    y_true = np.concatenate([np.ones(score_pos.shape[0]),
                             np.zeros(score_neg.shape[0])])
    y_score = np.concatenate([score_pos, score_neg])
    return dict(accuracy=correct / total, error=(total - correct) / total,
                auc=roc_auc_score(y_true, y_score))
