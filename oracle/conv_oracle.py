"""CPU oracle for the convolutional side of cfl (TEST INFRASTRUCTURE ONLY).

torch (CPU, float64, autograd) restatement of the reference's weight-normalised
convolution layers and of the ConvPCD encoder, following

    conv2d_weight_norm            cfl/layers.py:100-187
    conv2d_subpixel               cfl/layers.py:212-250   (== NHWC depth_to_space, block-major)
    conv2d_transpose_weight_norm  cfl/layers.py:253-361
    lrelu                         cfl/ops.py:10-12
    ConvPCD                       cfl/models/blocks.py:530-590
    FCPCD hidden fc_i layers      cfl/models/blocks.py:509-524, cfl/layers.py:28-97

PARITY: composition pinned, TensorFlow kernels unpinned (see oracle/cfl_oracle.py): the ConvPCD trunk + heads
+ losses + Adam reproduce, to float64 round-off, golden vectors produced by the reference's own CFL / ConvPCD
classes running over a TF stand-in (tests/golden/make_arith_goldens.py, tests/test_arith_goldens.py).  The
primitives follow the documented TF semantics of SURVEY.md App. E ('SAME' padding puts the extra pixel on the
bottom / right, l2_normalize(x, dims, eps=1e-12) = x * rsqrt(max(sum x^2, eps)), HWIO filters, NHWC
activations).  Only tests/ and smoke() may import it.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# Test hook: when set, every kinked activation of the oracles is evaluated as `x * slope` with the slope
# pattern supplied by the hook (called as MASK_HOOK(kind, x) -> bool mask of x's shape, in call order) instead of
# the sign of the oracle's own float64 pre-activation.  tests/test_activation_masks_gpu.py feeds the masks of the
# fp32 HIP run, which makes oracle and HIP the SAME locally-linear function.
MASK_HOOK = None


def lrelu(x, leak=0.2):
    if MASK_HOOK is not None:
        m = MASK_HOOK('lrelu', x)
        return x * torch.where(m, torch.ones_like(x), torch.full_like(x, leak))
    return torch.relu(x) - leak * torch.relu(-x)


def relu(x):
    if MASK_HOOK is not None:
        return x * MASK_HOOK('relu', x).to(x.dtype)
    return torch.relu(x)


def same_pads(n, k, s):
    """TF 'SAME': out = ceil(n/s); total pad = max((out-1)*s + k - n, 0); extra on the far side."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def wn_filter(V, g):
    """W = g * V / sqrt(max(sum_{h,w,i} V^2, 1e-12)) per output channel (cfl/layers.py:167-169)."""
    n2 = (V * V).sum(dim=(0, 1, 2), keepdim=True)
    W = V * torch.rsqrt(torch.clamp(n2, min=1e-12))
    return W * g.reshape(1, 1, 1, -1) if g is not None else W


def conv2d_weight_norm(x, V, g, b, stride, activation=None):
    """x: [B,H,W,Ci] NHWC; V: [KH,KW,Ci,Co] HWIO; 'SAME' padding."""
    KH, KW = V.shape[0], V.shape[1]
    _, pt, pb = same_pads(x.shape[1], KH, stride)
    _, pl, pr = same_pads(x.shape[2], KW, stride)
    W = wn_filter(V, g)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, W.permute(3, 2, 0, 1).contiguous(), stride=stride).permute(0, 2, 3, 1)
    if b is not None:
        y = y + b
    if activation == 'lrelu':
        y = lrelu(y)
    elif activation == 'relu':
        y = relu(y)
    return y


def conv2d_subpixel(x, scale=2, activation=None):
    """out[b, h*r+i, w*r+j, c] = in[b, h, w, (i*r+j)*C_out + c]."""
    B, H, W, C = x.shape
    co = C // (scale * scale)
    y = x.reshape(B, H, W, scale, scale, co).permute(0, 1, 3, 2, 4, 5).reshape(B, H * scale, W * scale, co)
    return relu(y) if activation == 'relu' else y


def xavier(rng, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def convpcd_layers(input_shape, dim=64, max_dim=512, min_dim=4):
    """[(Ci, Co)] of the 5x5 stride-2 layers and the flattened feature size."""
    h, w, c = input_shape
    start = min(h, w)
    layers = []
    ci = c
    while start % 2 == 0 and start > min_dim:
        start //= 2
        layers.append((ci, dim))
        ci = dim
        dim = min(dim * 2, max_dim)
        h, w = -(-h // 2), -(-w // 2)
    return layers, h * w * ci, (h, w, ci)


def init_convpcd(input_shape, rng, dtype=np.float64):
    layers, feat, _ = convpcd_layers(input_shape)
    p = {}
    for i, (ci, co) in enumerate(layers):
        p['conv%d/V' % (i + 1)] = xavier(rng, (5, 5, ci, co), 25 * ci, 25 * co).astype(dtype)
        p['conv%d/g' % (i + 1)] = np.ones(co, dtype)
        p['conv%d/b' % (i + 1)] = np.zeros(co, dtype)
    return p, feat


def fcpcd_hidden(x, p):
    """The hidden layers of FCPCD(layer_sizes=[...]) (cfl/models/blocks.py:516-524): fc_i = lrelu(fully_connected_weight_norm(x)),
    y = (x . V) * (g / sqrt(sum_rows V^2)) + b  (cfl/layers.py:80-94; no epsilon), parameters fc_<i>/{V [Ci, Co], g, b}."""
    y, i = x, 0
    while 'fc_%d/V' % i in p:
        V, g, b = p['fc_%d/V' % i], p['fc_%d/g' % i], p['fc_%d/b' % i]
        y = lrelu((y @ V) * (g / torch.sqrt((V * V).sum(dim=0))) + b)
        i += 1
    return y


def convpcd_features(x_flat, input_shape, p):
    """ConvPCD trunk: reshape NHWC -> [5x5 s2 wn-conv + lrelu]* -> flatten (cfl/models/blocks.py:569-589)."""
    y = x_flat.reshape((-1,) + tuple(input_shape))
    i = 1
    while 'conv%d/V' % i in p:
        y = conv2d_weight_norm(y, p['conv%d/V' % i], p['conv%d/g' % i], p['conv%d/b' % i], 2, 'lrelu')
        i += 1
    return y.reshape(y.shape[0], -1)
