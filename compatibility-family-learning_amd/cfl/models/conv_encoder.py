"""Encoder trunks on the GPU: weight-normalised layers in front of the pair-distance heads.

* ConvTrunk -- ConvPCD (cfl/models/blocks.py:530-590);
* FCTrunk   -- the hidden `fc_i` stack of FCPCD(layer_sizes=...) (cfl/models/blocks.py:509-527: weight-normalised fully
  connected layers with lrelu, zero-initialised biases, L2 term on V only), run as 1x1 convolutions on 1x1 images through the
  same cfl_conv2d_wn_{fwd,bwd} entry points (round 6: no command line of the reference sets layer_sizes, the constructor
  argument is kept for callers of the model classes).

ConvPCD trunk on the GPU (cfl/models/blocks.py:530-590): reshape to NHWC, then
[5x5 stride-2 weight-normalised conv + lrelu] while the side is even and > 4, flatten.
The flattened features feed the pair-distance heads (the fused HIP pair kernels with
D = flattened size); the trunk's gradient comes back through cfl_pair_input_grad and the
weight-normalised convolution backward (include/cfl_hip.h cfl_conv2d_wn_{fwd,bwd}).

Feature rows are kept in the order [pos_src, neg_src, pos_dst, neg_dst] so that each side's
2B rows are contiguous and no copy is needed between the trunk and the heads.
"""
import numpy as np
import torch

from .. import hipabi as H
from .base import xavier_uniform


def trunk_layers(input_shape, dim=64, max_dim=512, min_dim=4):
    h, w, c = input_shape
    start = min(h, w)
    layers, ci = [], c
    while start % 2 == 0 and start > min_dim:
        start //= 2
        layers.append((h, w, ci, dim))
        ci, dim = dim, min(dim * 2, max_dim)
        h, w = -(-h // 2), -(-w // 2)
    return layers, (h, w, ci)


class ConvTrunk(object):
    KSIZE, STRIDE = 5, 2                  # 5x5 stride-2 convolutions (cfl/models/blocks.py:571-586)
    SCOPE, FIRST = 'conv%d/Conv/', 1      # variable scopes conv1/Conv/{V,g,biases}, ...
    PREFIX = 'conv'                       # what this trunk's variables start with inside an encoder scope

    def _layers(self):
        return trunk_layers(self.input_shape)

    def _vshape(self, ci, co):
        return (self.KSIZE, self.KSIZE, ci, co)

    def __init__(self, input_shape, batch_rows, norm, reg_const, lr, beta1, beta2, eps, rng, device):
        self.input_shape = tuple(input_shape)
        self.device = device
        self.norm, self.reg_const = norm, float(reg_const or 0.0)
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.layers, out = self._layers()
        if not self.layers:
            raise ValueError('input shape %r leaves no trunk layer' % (self.input_shape,))
        self.feat_shape = out
        self.feature_size = out[0] * out[1] * out[2]
        if self.feature_size % 64:
            raise H.CflHipError('flattened trunk features (%d) must be a multiple of 64' % self.feature_size)
        k2 = self.KSIZE * self.KSIZE
        # flat parameter buffer [V1 g1 b1 V2 g2 b2 ...], each segment 64-float aligned
        self.slices, off = [], 0
        for (h, w, ci, co) in self.layers:
            seg = {}
            for name, n in (('V', k2 * ci * co), ('g', co), ('b', co)):
                seg[name] = (off, n)
                off += (n + 63) // 64 * 64
            self.slices.append(seg)
        self.theta = torch.zeros(off, dtype=torch.float32, device=device)
        for seg, (h, w, ci, co) in zip(self.slices, self.layers):
            # xavier_initializer: uniform +-sqrt(6 / (fan_in + fan_out)), fans include the receptive field
            V = np.asarray(rng.uniform(-1, 1, size=self._vshape(ci, co)) * np.sqrt(6.0 / (k2 * ci + k2 * co)),
                           np.float32)
            self._view(seg, 'V').copy_(torch.from_numpy(V).reshape(-1))
            self._view(seg, 'g').fill_(1.0)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.grad = torch.zeros_like(self.theta)
        self._bufs = {}

    def _view(self, seg, name, base=None):
        o, n = seg[name]
        return (self.theta if base is None else base)[o:o + n]

    def _plan(self, rows):
        plan = self._bufs.get(rows)
        if plan is None:
            convs, acts, wss = [], [], []
            for (h, w, ci, co) in self.layers:
                conv = H.make_conv(rows, h, w, ci, co, self.KSIZE, self.KSIZE, self.STRIDE, 'lrelu')
                oh, ow = H.conv_out_hw(conv)
                convs.append(conv)
                acts.append(torch.empty(rows, oh, ow, co, dtype=torch.float32, device=self.device))
                wss.append(H.conv_workspace(conv, self.device))
            plan = self._bufs[rows] = (convs, acts, wss)
        return plan

    def normalize(self, x):
        """data_normalizer (cfl/ops.py:66-124, scalar or per-channel) on the raw [rows, prod(shape)] pixels."""
        return self.norm.apply(x) if self.norm is not None else x.contiguous()

    def forward(self, x_rows):
        """x_rows: [rows, H*W*C] device tensor (already normalised) -> features [rows, F]."""
        rows = x_rows.shape[0]
        convs, acts, wss = self._plan(rows)
        cur = x_rows.reshape((rows,) + self.input_shape)
        self._inputs = [cur]
        for i, (conv, y, ws, seg) in enumerate(zip(convs, acts, wss, self.slices)):
            V = self._view(seg, 'V').view(self.KSIZE, self.KSIZE, conv.Ci, conv.Co)
            H.conv2d_wn_fwd(conv, cur, V, self._view(seg, 'g'), self._view(seg, 'b'), ws, y)
            cur = y
            self._inputs.append(cur)
        return cur.reshape(rows, self.feature_size)

    def backward(self, dfeat):
        """dfeat: [rows, F] -> fills self.grad (same layout as theta)."""
        rows = dfeat.shape[0]
        convs, acts, wss = self._plan(rows)
        dy = dfeat.reshape(acts[-1].shape)
        for i in reversed(range(len(convs))):
            conv, seg = convs[i], self.slices[i]
            V = self._view(seg, 'V').view(self.KSIZE, self.KSIZE, conv.Ci, conv.Co)
            dx, dV, dg, db = H.conv2d_wn_bwd(conv, self._inputs[i], V, self._view(seg, 'g'),
                                             self._inputs[i + 1], dy.contiguous(), wss[i],
                                             reg_const=self.reg_const, need_dx=i > 0, need_db=True)
            self._view(seg, 'V', self.grad).copy_(dV.reshape(-1))
            self._view(seg, 'g', self.grad).copy_(dg)
            self._view(seg, 'b', self.grad).copy_(db)
            dy = dx

    def apply_adam(self, lr_t, grad_scale=1.0):
        if self.reg_const:
            # s_loss_reg belongs to the weights the step was evaluated at: keep sum(V^2) of the pre-update filters
            # on the device (no host sync here)
            self._reg_sum = sum((self._view(seg, 'V') ** 2).sum() for seg in self.slices)
        H.adam_tf(self.theta, self.m, self.v, self.grad, lr_t, self.beta1, self.beta2, self.eps, grad_scale)

    def reg_loss(self):
        if not self.reg_const:
            return 0.0
        cached = getattr(self, '_reg_sum', None)
        if cached is not None:
            return 0.5 * self.reg_const * float(cached)
        tot = 0.0
        for seg in self.slices:
            tot += float((self._view(seg, 'V') ** 2).sum())
        return 0.5 * self.reg_const * tot

    def named(self, base=None):
        out = {}
        for i, (seg, (h, w, ci, co)) in enumerate(zip(self.slices, self.layers)):
            scope = self.SCOPE % (i + self.FIRST)
            out[scope + 'V'] = self._view(seg, 'V', base).view(self._vshape(ci, co)).cpu().numpy().copy()
            out[scope + 'g'] = self._view(seg, 'g', base).cpu().numpy().copy()
            out[scope + 'biases'] = self._view(seg, 'b', base).cpu().numpy().copy()
        return out

    def load_named(self, named, base=None):
        if base is None:
            self._reg_sum = None
        for i, seg in enumerate(self.slices):
            scope = self.SCOPE % (i + self.FIRST)
            for leaf, key in (('V', 'V'), ('g', 'g'), ('biases', 'b')):
                if scope + leaf in named:
                    self._view(seg, key, base).copy_(torch.as_tensor(
                        np.asarray(named[scope + leaf], np.float32)).reshape(-1))


class FCTrunk(ConvTrunk):
    """The hidden layers of FCPCD(layer_sizes=[...]) (cfl/models/blocks.py:509-527): `fc_i` = weight-normalised fully connected
    layer + lrelu (cfl/layers.py:28-97: y = (x . V) * g / ||V_col|| + b), variables fc_<i>/fully_connected/{V [Ci, Co], g, biases},
    L2 term on V only (weights_regularizer; the biases have none).  A fully connected layer IS a 1x1 convolution on a 1x1
    image: the layers run through the same weight-normalised convolution kernels as the conv trunk, rows as the batch."""
    KSIZE, STRIDE = 1, 1
    SCOPE, FIRST = 'fc_%d/fully_connected/', 0
    PREFIX = 'fc_'

    def __init__(self, input_size, layer_sizes, batch_rows, norm, reg_const, lr, beta1, beta2, eps, rng, device):
        self.layer_sizes = [int(n) for n in layer_sizes]
        if not self.layer_sizes or min(self.layer_sizes) <= 0:
            raise ValueError('layer_sizes %r' % (layer_sizes,))
        ConvTrunk.__init__(self, (1, 1, int(input_size)), batch_rows, norm, reg_const, lr, beta1, beta2, eps, rng, device)

    def _layers(self):
        layers, ci = [], self.input_shape[2]
        for co in self.layer_sizes:
            layers.append((1, 1, ci, co))
            ci = co
        return layers, (1, 1, ci)

    def _vshape(self, ci, co):
        return (ci, co)
