"""MrCGAN generator / discriminator stacks on the GPU with hand-written backward passes.

    SRGenerator             cfl/models/blocks.py:25-109
    SRDiscriminator         cfl/models/blocks.py:112-247
    ConvTransposeGenerator  cfl/models/blocks.py:250-332
    ConvDiscriminator       cfl/models/blocks.py:335-438

Every layer is a call into libcfl_hip.so (weight-norm conv / transposed conv / FC on fp32
MFMA, element-wise and permutation kernels of csrc/cfl_gan.hip); torch only owns the buffers.
Each network keeps its variables, Adam slots and gradients in ONE flat fp32 array each, so the
optimiser is a single cfl_adam_tf launch.  There is no autograd: a forward pass records a tape
of (layer, input, output) and the backward walks it; the gradient-penalty term needs the
double backward of the discriminator, which for this piecewise-linear network is a second,
bias-free forward pass with the activation slopes frozen (see `gp_grads`).
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import hipabi as H
from .. import hipgan as G


def up_count(shape, min_dim=4):
    start = min(shape[0], shape[1])
    n = 0
    while start % 2 == 0 and start > min_dim:
        start //= 2
        n += 1
    return n, start


def xavier(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return np.asarray(rng.uniform(-lim, lim, size=shape), np.float32)


class ParamPool(object):
    """Flat fp32 storage of a network's variables (theta), Adam slots (m, v), gradient (grad)
    and a second gradient buffer (grad2: the gradient-penalty term before it is added)."""

    def __init__(self, device):
        self.device = device
        self.specs = {}      # name -> (offset, shape)
        self.order = []
        self._init = {}
        self.total = 0
        self.theta = None
        self.version = 0     # bumped whenever theta changes (Adam, load): per-layer weight caches key on it
        self.layers = []     # the WNLayers whose variables live here (cache preparation behind the optimizer step)
        self._views = {}     # id(buffer) -> (weak reference to the buffer, {name: view})

    def add(self, name, value):
        value = np.asarray(value, np.float32)
        self.specs[name] = (self.total, value.shape)
        self.order.append(name)
        self._init[name] = value
        self.total += (value.size + 63) // 64 * 64

    def finalize(self):
        host = np.zeros(self.total, np.float32)
        for n in self.order:
            o, shp = self.specs[n]
            host[o:o + int(np.prod(shp))] = self._init[n].reshape(-1)
        self.theta = torch.from_numpy(host).to(self.device)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.grad = torch.zeros_like(self.theta)
        self.grad2 = torch.zeros_like(self.theta)
        self._init = None

    def view(self, name, base=None):
        """the variable `name` as a view of `base` (theta, an Adam slot or a gradient buffer).  Views are cached per
        (name, buffer): a post-epoch step asks for ~600 of them, and slicing + reshaping each time was a fifth of the
        host time of the step (tools/gan_host_profile.py)."""
        base = self.theta if base is None else base
        # keyed by the buffer OBJECT: an address says nothing about dtype / length, and a cache keyed by it would keep
        # every buffer that was ever passed alive.  The pool's own buffers (theta, m, v, grad, grad2) live as long as the pool;
        # anything else -- a caller's scratch buffer -- is evicted when a new buffer object shows up.
        if base.dtype != torch.float32 or base.numel() < self.total:
            raise ValueError('ParamPool.view: buffer of %d %s elements for a pool of %d floats' % (base.numel(), base.dtype, self.total))
        bid = id(base)
        cache = self._views.get(bid)
        if cache is None or cache[0]() is not base:
            own = {id(b) for b in (self.theta, getattr(self, 'm', None), getattr(self, 'v', None), getattr(self, 'grad', None),
                                   getattr(self, 'grad2', None)) if b is not None}
            for k in [k for k, c in self._views.items() if k not in own or c[0]() is None]:
                del self._views[k]
            import weakref
            cache = self._views[bid] = (weakref.ref(base), {})
        v = cache[1].get(name)
        if v is None:
            o, shp = self.specs[name]
            v = cache[1][name] = base[o:o + int(np.prod(shp))].view(shp)
        return v

    def named(self, base=None):
        return {n: self.view(n, base).detach().cpu().numpy().copy() for n in self.order}

    def touch(self):
        """theta was modified: the layers' cached weight-norm scales / prepared filters are stale"""
        self.version += 1

    def load(self, values, base=None):
        if base is None:
            self.touch()
        for n in self.order:
            if n in values:
                self.view(n, base).copy_(torch.as_tensor(np.asarray(values[n], np.float32)).to(self.device))


class WNLayer(object):
    """One weight-normalised layer: kind 'conv' (V [KH,KW,Ci,Co]), 'convt' (V [KH,KW,Co,Ci]) or
    'fc' (V [Ci,Co], run as a 1x1 convolution on a 1x1 image)."""

    def __init__(self, pool, scope, kind, kh, kw, ci, co, stride, act, rng):
        self.pool, self.scope, self.kind = pool, scope, kind
        self.kh, self.kw, self.ci, self.co, self.stride, self.act = kh, kw, ci, co, stride, act
        if kind == 'convt':
            V = xavier(rng, (kh, kw, co, ci), kh * kw * co, kh * kw * ci)
        elif kind == 'conv':
            V = xavier(rng, (kh, kw, ci, co), kh * kw * ci, kh * kw * co)
        else:
            V = xavier(rng, (ci, co), ci, co)
        pool.add(scope + '/V', V)
        pool.add(scope + '/g', np.ones(co, np.float32))
        pool.add(scope + '/biases', np.zeros(co, np.float32))
        self._desc = {}
        # what depends on the weights only (weight-norm scale, prepared filter planes) is kept across the 3-6 calls the
        # layer sees per step and rebuilt when the pool's version moves (include/cfl_hip.h: cfl_conv2d_wn_*_cached)
        self._cache = G.ConvCache() if kind != 'convt' else None
        self._cache_version = -1
        self._prep_desc = None       # shape of the layer's first forward call: what `prepare` rebuilds the cache for
        self._slabs = {}             # (workspace, call shape) -> split-K slab region of the deferred weight-gradient finalisation
        pool.layers.append(self)

    def cache(self):
        if self._cache is not None and self._cache_version != self.pool.version:
            self._cache.invalidate()
            self._cache_version = self.pool.version
        return self._cache

    def p(self, what, base=None):
        return self.pool.view(self.scope + '/' + what, base)

    def prepare(self):
        """rebuild the cache (scale, filter planes) for the current weights on the current stream, without running the layer"""
        if self._cache is not None and self._prep_desc is not None:
            G.conv_prepare(self._prep_desc, self.p('V'), self.p('g'), self.cache())

    def desc(self, B, Hh, W, act):
        key = (B, Hh, W, act)
        d = self._desc.get(key)
        if d is None:
            d = self._desc[key] = H.CflConv(B, Hh, W, self.ci, self.co, self.kh, self.kw, self.stride,
                                            H.CONV_ACTS[act])
        return d

    def out_hw(self, Hh, W):
        if self.kind == 'convt':
            return Hh * self.stride, W * self.stride
        return -(-Hh // self.stride), -(-W // self.stride)

    def fwd(self, x, ws, act='layer', bias=True, residual=None, subpixel=False):
        """x [B,H,W,Ci] -> y [B,OH,OW,Co].  residual [B,OH,OW,Co]: y = act(conv + b + residual) (a residual block's join as
        the store epilogue of its second convolution); subpixel: y is stored 2x sub-pixel shuffled, [B,2OH,2OW,Co/4] -- the
        activation (element-wise) then plays the role of the shuffle's own."""
        act = self.act if act == 'layer' else act
        B, Hh, W, _ = x.shape
        d = self.desc(B, Hh, W, act)
        if self._prep_desc is None:
            self._prep_desc = d
        t = self.kind == 'convt'
        oh, ow = self.out_hw(Hh, W)
        shape = (B, 2 * oh, 2 * ow, self.co // 4) if subpixel else (B, oh, ow, self.co)
        y = torch.empty(shape, dtype=torch.float32, device=x.device)
        G.conv_fwd(d, x, self.p('V'), self.p('g'), self.p('biases') if bias else None, y, ws.get(d, t), t, cache=self.cache(),
                   residual=residual, subpixel=subpixel)
        return y

    def bwd(self, x, y, dy, ws, need_dx=True, need_dw=True, grad=None, need_db=True, act='layer', dy_subpixel=False):
        """dx (or None); parameter gradients are written into `grad` (a flat buffer shaped like theta).
        `act` overrides the activation whose slope at `y` multiplies dy (a residual join's lrelu folded into the
        backward of the block's second convolution: y = the join's output).

        The weight-gradient product (GEMM, split-K sums, weight-norm finalisation: 3-5 launches) does not feed the
        backward chain -- only dx does -- so it is enqueued on the workspace's SIDE stream and overlaps the input
        gradients of this and the following layers (round 4: most launches of the MrCGAN step are far too small to
        fill 256 CUs on their own).  `Workspace.join()` orders the side stream back before the gradients are read."""
        t = self.kind == 'convt'
        act = self.act if act == 'layer' else act
        B, Hh, W, _ = x.shape
        d = self.desc(B, Hh, W, act)
        dx = torch.empty(B, Hh, W, self.ci, dtype=torch.float32, device=dy.device) if need_dx else None
        cache = self.cache()
        dV = self.p('V', grad) if need_dw else None
        dg = self.p('g', grad) if need_dw else None
        db = self.p('biases', grad) if (need_dw and need_db) else None
        side = ws.side_stream if (need_dw and not t and cache is not None and G.ConvCache.enabled
                                  and (cache.flags.value & 1)) else None      # (bit 0: the cached weight-norm scale is valid)
        sp = dict(dy_subpixel=True) if dy_subpixel else {}      # dy / y in the shuffled layout of fwd(subpixel=True)
        if side is None:
            G.conv_bwd(d, x, self.p('V'), self.p('g'), y, dy, ws.get(d, t), dx=dx, dV=dV, dg=dg, db=db, transposed=t,
                       cache=cache, **sp)
            return dx
        side.wait_stream(torch.cuda.current_stream())          # x, y, dy were produced on the main stream
        with torch.cuda.stream(side):
            if Workspace.defer_wfinal:
                # only the contraction now, into this layer's own slab region; the slab sums and the weight-norm finalisation of
                # ALL the layers of this backward chain run as one launch pair at Workspace.join() (cfl_conv_wfinal_many:
                # ~80 launches of 5-16 us per MrCGAN step before)
                slab = self._slab(ws, d)
                G.conv_wgrad_slabs(d, x, y, dy, slab, dy_subpixel=dy_subpixel)
                ws.pending.append((d, slab, self.p('V'), self.p('g'), cache, dV, dg, db))
                if len(ws.pending) >= Workspace.defer_group:
                    # finished in groups along the chain, not all at its end: a backward pass meets the wide layers first, and one
                    # launch for all 23 layers at the join put their 0.2 ms of per-channel walks on the chain's tail (+0.4 ms per step)
                    G.conv_wfinal_many(ws.pending)
                    ws.pending = []
            else:
                G.conv_bwd(d, x, self.p('V'), self.p('g'), y, dy, ws.side_ws.get(d, t), dx=None, dV=dV, dg=dg, db=db,
                           transposed=t, cache=cache, **sp)
        for tt in (x, y, dy):
            if tt is not None:
                tt.record_stream(side)                          # the allocator must not recycle them under the side stream
        ws.side_used = True
        if need_dx:
            G.conv_bwd(d, x, self.p('V'), self.p('g'), y, dy, ws.get(d, t), dx=dx, dV=None, dg=None, db=None,
                       transposed=t, cache=cache, **sp)
        return dx

    def _slab(self, ws, d):
        """this layer's split-K slab region for weight gradients taken through workspace `ws` at call shape `d` (kept: a layer is
        differentiated by up to three backward chains per step, each with its own workspace, on its own stream)"""
        key = (id(ws), d.B, d.H, d.W, d.act)
        slab = self._slabs.get(key)
        if slab is None:
            slab = self._slabs[key] = torch.empty(G.conv_wgrad_slab_bytes(d) // 4, dtype=torch.float32, device=self.pool.device)
        return slab

    def bwd_takes_subpixel(self, B, Hh, W):
        """can bwd(..., dy_subpixel=True) be used for an input of this shape (see hipgan.conv_bwd_takes_subpixel)?"""
        return self.kind == 'conv' and G.conv_bwd_takes_subpixel(self.desc(B, Hh, W, None))


class Workspace(object):
    # CFL_GAN_OVERLAP=0 keeps every launch of a network on one stream (A/B runs, debugging)
    overlap = os.environ.get('CFL_GAN_OVERLAP', '1') not in ('0', '')
    # CFL_GAN_DEFER_WFINAL=1 (opt-in): the slab sums + weight-norm finalisations of a backward chain's layers in grouped launch pairs
    # (cfl_conv_wfinal_many) instead of a pair per layer.  Built and measured in round 6 (tools/r06_gan_ab.sh, stream tuner off, same
    # box, three alternations): 83 launches fewer per step but the step is 0.2-0.4 ms SLOWER (12.2-12.3 -> 12.4-12.7 ms, groups of
    # 3 ... all layers) -- the per-layer finalisations fill gaps of the side stream as they come, a grouped launch serialises the
    # wide layers' per-channel walks behind the chain.  The step is GPU-bound: launch count is not what it pays for.  Default off.
    defer_wfinal = os.environ.get('CFL_GAN_DEFER_WFINAL', '0') not in ('0', '')
    defer_group = int(os.environ.get('CFL_GAN_DEFER_GROUP', '6'))     # layers per finalisation launch pair

    def __init__(self, device, side=True):
        self.device = device
        self.buf = torch.empty(1024, dtype=torch.float32, device=device)
        self._need = {}
        # weight-gradient products run on a side stream with a workspace of their own (WNLayer.bwd)
        self.side_stream = torch.cuda.Stream(device=device) if (side and Workspace.overlap) else None
        self.side_ws = Workspace(device, side=False) if self.side_stream is not None else None
        self.side_used = False
        self.pending = []            # deferred weight-gradient finalisations of the backward chain in flight (WNLayer.bwd)

    def join(self):
        """finish the deferred weight gradients (one launch pair for all layers of the chain) and order the side stream's work
        before whatever the current stream does next"""
        if self.side_stream is not None and self.side_used:
            if self.pending:
                with torch.cuda.stream(self.side_stream):
                    G.conv_wfinal_many(self.pending)
                self.pending = []
            torch.cuda.current_stream().wait_stream(self.side_stream)
            self.side_used = False

    def get(self, d, transposed):
        key = (d.B, d.H, d.W, d.Ci, d.Co, d.KH, d.KW, d.stride, transposed)
        n = self._need.get(key)
        if n is None:
            n = self._need[key] = (G.conv_ws_bytes(d, transposed) + 3) // 4
        if self.buf.numel() < n:
            self.buf = torch.empty(n, dtype=torch.float32, device=self.device)
        return self.buf


class _Net(object):
    def __init__(self, device, lr, beta1, beta2, eps=1e-8):
        self.device = device
        self.pool = ParamPool(device)
        self.ws = Workspace(device)
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.beta1_power, self.beta2_power = np.float32(beta1), np.float32(beta2)

    def lr_t(self):
        return float(np.float32(self.lr) * np.sqrt(np.float32(1) - self.beta2_power) /
                     (np.float32(1) - self.beta1_power))

    # CFL_GAN_PREP_AHEAD=0: leave every layer's cache to its first convolution of the next step (A/B runs)
    prep_ahead = os.environ.get('CFL_GAN_PREP_AHEAD', '1') not in ('0', '')
    # CFL_GAN_PREP_BATCHED=0: one preparation call per layer (the form of round 5; A/B runs)
    prep_batched = os.environ.get('CFL_GAN_PREP_BATCHED', '1') not in ('0', '')

    def prepare_caches(self):
        """Right behind the optimizer step: rebuild every layer's weight-norm scale and filter planes on a side stream, so
        that the next step's forward chains do not carry ~3 small launches in front of every layer's first convolution
        (`join_prepare` orders the side stream back in front of the first use)."""
        if not (_Net.prep_ahead and Workspace.overlap and G.ConvCache.enabled):
            return
        if not hasattr(self, '_prep_stream'):
            self._prep_stream = torch.cuda.Stream(device=self.device)
        self._prep_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._prep_stream):
            complete = True
            many = []
            for layer in self.pool.layers:
                complete = complete and (layer._cache is None or layer._prep_desc is not None)
                if layer._cache is not None and layer._prep_desc is not None:
                    many.append((layer._prep_desc, layer.p('V'), layer.p('g'), layer.cache()))
            # every layer's weight-norm scale in ONE launch, every layer's filter planes in another (round 6: ~65 launches in a
            # row before, which the next step's first layers waited for)
            if _Net.prep_batched:
                G.conv_prepare_many(many)
            else:
                for layer in self.pool.layers:
                    layer.prepare()
            self._prep_event = torch.cuda.Event()
            self._prep_event.record()
        # every cache was rebuilt for THIS version of the weights behind `_prep_event` (caches_prepared)
        self._prepared_version = self.pool.version if complete else -1

    def caches_prepared(self):
        """Were ALL layer caches rebuilt ahead (prepare_caches) for the current weights?  Then their host validity bits are
        set and every kernel that filled them sits behind `_prep_event`, which every consumer waits for on its own stream
        (join_prepare): forward passes may run on several streams at once.  Otherwise the first stream to touch a layer
        builds its cache and sets the HOST bits at once -- a second stream would read scale / planes with no dependency on
        the kernels that write them -- so concurrent forwards are only allowed when this returns True (GanPhase.step)."""
        if not (_Net.prep_ahead and Workspace.overlap and G.ConvCache.enabled):
            return False
        if getattr(self, '_prep_event', None) is None or getattr(self, '_prepared_version', -1) != self.pool.version:
            return False
        return all(l._cache is None or (l._cache_version == self.pool.version and l._cache.flags.value != 0)
                   for l in self.pool.layers)

    def join_prepare(self):
        """every entry point that reads a layer's cache waits -- on whatever stream it runs -- for the last preparation"""
        ev = getattr(self, '_prep_event', None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def adam(self, grad_scale=1.0):
        p = self.pool
        H.adam_tf(p.theta, p.m, p.v, p.grad, self.lr_t(), self.beta1, self.beta2, self.eps, grad_scale)
        p.touch()
        self.prepare_caches()
        self.beta1_power = np.float32(self.beta1_power * np.float32(self.beta1))
        self.beta2_power = np.float32(self.beta2_power * np.float32(self.beta2))

    def state(self):
        return {'variables': self.pool.named(), 'adam_m': self.pool.named(self.pool.m),
                'adam_v': self.pool.named(self.pool.v), 'beta1_power': float(self.beta1_power),
                'beta2_power': float(self.beta2_power)}

    def load_state(self, st):
        self.pool.load(st['variables'])
        self.pool.load(st.get('adam_m', {}), self.pool.m)
        self.pool.load(st.get('adam_v', {}), self.pool.v)
        self.beta1_power = np.float32(st.get('beta1_power', self.beta1))
        self.beta2_power = np.float32(st.get('beta2_power', self.beta2))


class Generator(_Net):
    """gan_type 'srgan' = SRGenerator, 'conv' = ConvTransposeGenerator.  Input: concat(z, c)."""
    # CFL_GAN_FUSE_UNSHUFFLE=1: leave the un-shuffle of a sub-pixel stage's gradient to the dy loaders of the convolution below
    # it (cfl_conv2d_wn_bwd_fused) instead of the separate pass.  Bit-identical and OPT-IN: measured 0.3-0.5 ms SLOWER per
    # MrCGAN step (14.0-14.1 -> 14.3-14.6 ms, three alternations on one box) -- the pass it removes runs at memory speed, while
    # the loaders of the two matrix-bound kernels then read every second pixel of the shuffled tensor and its activation too
    fuse_unshuffle = os.environ.get('CFL_GAN_FUSE_UNSHUFFLE', '0') not in ('0', '')

    def __init__(self, gan_type, ae_shape, in_dim, data_type, rng, device, lr=2e-4, beta1=0.5, beta2=0.999,
                 dim=64, scope='Generator', c_dim=None, t_dim=None):
        """in_dim = z_dim + c_dim.  With t_dim (cgan --t-dim) the last c_dim input columns first go through
        the weight-norm layer fc_t (lrelu) (cfl/models/blocks.py:55-64)."""
        super(Generator, self).__init__(device, lr, beta1, beta2)
        self.gan_type, self.ae_shape, self.data_type, self.dim = gan_type, tuple(ae_shape), data_type, dim
        nb, start = up_count(self.ae_shape)
        self.nb, self.start = nb, start
        ch = self.ae_shape[2]
        self.blocks = []
        self.fc_t = None
        self.c_dim, self.z_dim = c_dim, (in_dim - c_dim if c_dim else None)
        if t_dim:
            self.fc_t = WNLayer(self.pool, scope + '/fc_t/fully_connected', 'fc', 1, 1, c_dim, t_dim, 1, 'lrelu', rng)
            in_dim = in_dim - c_dim + t_dim
        if gan_type == 'srgan':
            self.fc_dim = dim
            self.fc1 = WNLayer(self.pool, scope + '/fc1/fully_connected', 'fc', 1, 1, in_dim, dim * start * start, 1,
                               'relu', rng)
            ci = dim
            for i in range(nb - 1):
                co = 4 * dim * (2 ** (nb - i - 1))
                self.blocks.append(WNLayer(self.pool, scope + '/subpixel_block%d/Conv' % (i + 1), 'conv', 3, 3, ci,
                                           co, 1, None, rng))
                ci = co // 4
            self.out = WNLayer(self.pool, scope + '/outputs/Conv', 'conv', 3, 3, ci, 4 * ch, 1, None, rng)
        elif gan_type == 'conv':
            scale = 2 ** (nb - 1)
            self.fc_dim = dim * scale
            self.fc1 = WNLayer(self.pool, scope + '/fc1/fully_connected', 'fc', 1, 1, in_dim,
                               dim * scale * start * start, 1, 'relu', rng)
            ci = dim * scale
            for i in range(nb - 1):
                co = dim * (2 ** (nb - i - 1))
                self.blocks.append(WNLayer(self.pool, scope + '/conv_t%d/Conv2d_transpose' % (i + 1), 'convt', 5, 5,
                                           ci, co, 2, 'relu', rng))
                ci = co
            self.out = WNLayer(self.pool, scope + '/outputs/Conv2d_transpose', 'convt', 5, 5, ci, ch, 2, None, rng)
        else:
            raise ValueError('unknown gan_type %r' % (gan_type,))
        self.pool.finalize()

    def forward(self, zc):
        """zc [N, z_dim + c_dim] -> (activations [N, prod(ae_shape)], tape)."""
        self.join_prepare()
        N = zc.shape[0]
        tape = []
        if self.fc_t is not None:
            c = torch.empty(N, self.c_dim, dtype=torch.float32, device=self.device)
            G.copy_cols(zc, self.z_dim, c, 0, self.c_dim)
            ct = self.fc_t.fwd(c.view(N, 1, 1, -1), self.ws)
            tape.append(('fc_t', c.view(N, 1, 1, -1), ct))
            zt = torch.empty(N, self.z_dim + self.fc_t.co, dtype=torch.float32, device=self.device)
            G.copy_cols(zc, 0, zt, 0, self.z_dim)
            G.copy_cols(ct.view(N, -1), 0, zt, self.z_dim, self.fc_t.co)
            zc = zt
        x = zc.view(N, 1, 1, -1)
        h = self.fc1.fwd(x, self.ws)
        tape.append(('layer', self.fc1, x, h))
        h = h.view(N, self.start, self.start, self.fc_dim)
        sr = self.gan_type == 'srgan'
        for blk in self.blocks:
            if sr:
                # conv -> sub-pixel shuffle -> relu in ONE launch: the shuffle is the convolution's store epilogue and the
                # un-shuffled output, which nothing reads (the layer has no activation of its own), is never materialised
                s = blk.fwd(h, self.ws, act='relu', subpixel=True)
                tape.append(('layer', blk, h, None))
                tape.append(('subpixel', 'relu', s))
                y = s
            else:
                y = blk.fwd(h, self.ws)
                tape.append(('layer', blk, h, y))
            h = y
        if sr:
            y = self.out.fwd(h, self.ws, subpixel=True)
            tape.append(('layer', self.out, h, None))
            tape.append(('subpixel', None, None))
        else:
            y = self.out.fwd(h, self.ws)
            tape.append(('layer', self.out, h, y))
        outputs = y.view(N, -1)
        acts = G.act_fwd(outputs, self.data_type) if self.data_type != 'linear' else outputs
        tape.append(('data_act', acts))
        return acts, tape

    def backward(self, tape, d_acts):
        """Writes d g_total / d generator variables into pool.grad (d_acts = d g_total / d activations)."""
        self.join_prepare()     # (a no-op after a forward; a backward over an old tape right behind an optimizer step must wait too)
        N = d_acts.shape[0]
        acts = tape[-1][1]
        d = G.act_bwd(acts, d_acts, self.data_type) if self.data_type != 'linear' else d_acts
        shuffled = None      # (act, s) of a sub-pixel stage whose un-shuffle is left to the dy loaders of the layer below it
        for idx in range(len(tape) - 2, -1, -1):
            item = tape[idx]
            if item[0] == 'subpixel':
                _, act, s = item
                shape = s.shape if s is not None else (N,) + self.ae_shape
                d = d.reshape(shape)
                # the convolution that feeds this shuffle is the tape item in front of it
                layer, x = tape[idx - 1][1], tape[idx - 1][2]
                if Generator.fuse_unshuffle and layer.bwd_takes_subpixel(x.shape[0], x.shape[1], x.shape[2]):
                    shuffled = (act, s)
                else:
                    d = G.subpixel_bwd(s, d, act)
            elif item[0] == 'fc_t':
                _, c, ct = item      # d = d loss / d [z | fc_t(c)]: only the fc_t columns carry on
                dct = torch.empty(N, self.fc_t.co, dtype=torch.float32, device=self.device)
                G.copy_cols(d.reshape(N, -1), self.z_dim, dct, 0, self.fc_t.co)
                self.fc_t.bwd(c, ct, dct.view(ct.shape), self.ws, need_dx=False, need_dw=True, grad=self.pool.grad)
            else:
                _, layer, x, y = item
                need_dx = layer is not self.fc1 or self.fc_t is not None
                N_, Hh, W = x.shape[0], x.shape[1], x.shape[2]
                oh, ow = layer.out_hw(Hh, W)
                if shuffled is not None:
                    # d is still [N, 2oh, 2ow, co/4]; the shuffle's activation slope (at s) is applied by the loaders too
                    act, s = shuffled
                    shuffled = None
                    d = layer.bwd(x, s, d, self.ws, need_dx=need_dx, need_dw=True, grad=self.pool.grad, act=act, dy_subpixel=True)
                else:
                    d = layer.bwd(x, y, d.reshape(N_, oh, ow, layer.co), self.ws, need_dx=need_dx, need_dw=True, grad=self.pool.grad)
        self.ws.join()


class Discriminator(_Net):
    """gan_type 'srgan' = SRDiscriminator, 'conv' = ConvDiscriminator; heads disc_outputs (1 logit)
    and latent_outputs (latent_size)."""

    def __init__(self, gan_type, ae_shape, latent_size, rng, device, lr=2e-4, beta1=0.5, beta2=0.999,
                 scope='Discriminator', c_dim=None, t_dim=None):
        """c_dim: width of the cgan condition t, tiled over the feature map and concatenated in front of
        the down-sampling conv of stage 3 (srgan, cfl/models/blocks.py:182-195; images >= 64 pixels only) or
        after conv(nb-1) (conv, :382-395); with t_dim it first goes through <stage>/fc_t (lrelu)."""
        super(Discriminator, self).__init__(device, lr, beta1, beta2)
        self.gan_type, self.ae_shape, self.latent_size = gan_type, tuple(ae_shape), latent_size
        nb, _ = up_count(self.ae_shape)
        h, w, ch = self.ae_shape
        self.stages = []   # [(res conv pairs, down conv)]
        self.cond_stage, self.fc_t, self.t_channels = None, None, 0

        def cond(stage_scope):
            self.t_channels = t_dim or c_dim
            if t_dim:
                self.fc_t = WNLayer(self.pool, stage_scope + 'fc_t/fully_connected', 'fc', 1, 1, c_dim, t_dim, 1,
                                    'lrelu', rng)
            return self.t_channels
        if gan_type == 'srgan':
            dim = 32
            self.stem = WNLayer(self.pool, scope + '/conv/Conv', 'conv', 4, 4, ch, dim, 2, 'lrelu', rng)
            h, w = -(-h // 2), -(-w // 2)
            for i in range(nb):
                s = scope + '/conv%d/' % (i + 1)
                cn = lambda j: 'Conv' if j == 0 else 'Conv_%d' % j
                res = []
                for j in range(2):
                    a = WNLayer(self.pool, s + cn(2 * j), 'conv', 3, 3, dim, dim, 1, 'lrelu', rng)
                    b = WNLayer(self.pool, s + cn(2 * j + 1), 'conv', 3, 3, dim, dim, 1, None, rng)
                    res.append((a, b))
                extra = 0
                if c_dim and i == 3:
                    self.cond_stage = i          # concat BEFORE this stage's down conv
                    extra = cond(s)
                down = WNLayer(self.pool, s + cn(4), 'conv', 4, 4, dim + extra, dim * 2, 2, 'lrelu', rng)
                self.stages.append((res, down))
                dim *= 2
                h, w = -(-h // 2), -(-w // 2)
            feat = h * w * dim
        elif gan_type == 'conv':
            dim, ci = 64, ch
            self.stem = None
            for i in range(nb):
                down = WNLayer(self.pool, scope + '/conv%d/Conv' % (i + 1), 'conv', 5, 5, ci, dim, 2, 'lrelu', rng)
                self.stages.append(([], down))
                ci, dim = dim, min(dim * 2, 512)
                if c_dim and i == nb - 2:
                    self.cond_stage = i + 1      # concat AFTER conv(i+1) = before the next stage's conv
                    ci += cond(scope + '/conv%d/' % (i + 1))
                h, w = -(-h // 2), -(-w // 2)
            feat = h * w * ci
        else:
            raise ValueError('unknown gan_type %r' % (gan_type,))
        self.feat = feat
        self.disc_head = WNLayer(self.pool, scope + '/disc_outputs/fully_connected', 'fc', 1, 1, feat, 1, 1, None, rng)
        self.lat_head = WNLayer(self.pool, scope + '/latent_outputs/fully_connected', 'fc', 1, 1, feat, latent_size,
                                1, None, rng)
        self.pool.finalize()

    # tape entries: ('conv', layer, x, y) | ('res', a, b, h, r1, r2, out) | ('tcat', C1, t_in, t_out)
    def forward(self, x_flat, t=None, ws=None):
        """`ws`: the workspace to use (a chain's own when a second forward of this network runs on another stream)"""
        self.join_prepare()
        N = x_flat.shape[0]
        tape = []
        h = x_flat.view((N,) + self.ae_shape)
        if ws is None:
            return self._forward(h, N, t, self.ws, tape)
        return self._forward(h, N, t, ws, tape)

    def _forward(self, h, N, t, ws, tape):
        if self.stem is not None:
            y = self.stem.fwd(h, ws)
            tape.append(('conv', self.stem, h, y))
            h = y
        for si, (res, down) in enumerate(self.stages):
            for a, b in res:
                r1 = a.fwd(h, ws)
                # out = lrelu(conv_b(r1) + h): the join is the store epilogue of conv b (r2 itself is read by nothing)
                out = b.fwd(r1, ws, act='lrelu', residual=h)
                tape.append(('res', a, b, h, r1, None, out))
                h = out
            if self.cond_stage == si:
                if t is None:
                    raise H.CflHipError('this discriminator is conditional: pass t')
                t_in = t.contiguous()
                t_out = self.fc_t.fwd(t_in.view(N, 1, 1, -1), ws).view(N, -1) if self.fc_t is not None else t_in
                C1 = h.shape[3]
                h = G.tile_concat_channels(h, t_out)
                tape.append(('tcat', C1, t_in, t_out))
            y = down.fwd(h, ws)
            tape.append(('conv', down, h, y))
            h = y
        f = h.view(N, 1, 1, self.feat)
        disc = self.disc_head.fwd(f, ws).view(N, 1)
        lat = self.lat_head.fwd(f, ws).view(N, self.latent_size)
        return disc, lat, (tape, f)

    def chain(self, k):
        """(stream, workspace) of independent backward chain k = 1, 2 (the post-epoch step runs the d-loss backward, the
        gradient-penalty passes and the g-loss backward + generator backward side by side: cfl/models/mrcgan.py); None
        when CFL_GAN_OVERLAP=0"""
        if not Workspace.overlap:
            return None
        if not hasattr(self, '_chains'):
            self._chains = [(torch.cuda.Stream(device=self.device), Workspace(self.device)) for _ in range(2)]
        return self._chains[k - 1]

    def backward(self, tapef, lo, hi, d_disc, d_lat, need_dx, need_dw, grad=None, record=None, ws=None):
        """Backward over the rows [lo, hi) of a recorded forward.  d_disc [n,1] / d_lat [n,L] (either
        may be None).  Returns d/d input rows [n, prod(ae_shape)] when need_dx.  `record`, when a
        list, receives the upstream gradient of every layer (for `gp_grads`).  `ws`: the workspace to use (a chain's own
        when several backward passes of this network run on different streams)."""
        self.join_prepare()
        tape, f = tapef
        n = hi - lo
        ws = self.ws if ws is None else ws
        grad = self.pool.grad if grad is None else grad
        fr = f[lo:hi]
        df = None
        if d_disc is not None:
            dy = d_disc.view(n, 1, 1, 1)
            if record is not None:
                record.append(dy)
            df = self.disc_head.bwd(fr, None, dy, ws, True, need_dw, grad)
        elif need_dw:
            for w in ('V', 'g', 'biases'):
                self.disc_head.p(w, grad).zero_()
        if d_lat is not None:
            dl = self.lat_head.bwd(fr, None, d_lat.view(n, 1, 1, self.latent_size), ws, True, need_dw, grad)
            df = dl if df is None else G.axpy(1.0, dl, df)
        elif need_dw:
            for w in ('V', 'g', 'biases'):
                self.lat_head.p(w, grad).zero_()
        first_layer = tape[0][1]
        d = None
        for item in reversed(tape):
            if item[0] == 'tcat':
                _, C1, t_in, t_out = item
                dh, dt = G.tile_concat_channels_bwd(d.contiguous(), C1, True, need_dw and self.fc_t is not None)
                if dt is not None:
                    self.fc_t.bwd(t_in[lo:hi].view(n, 1, 1, -1), t_out[lo:hi].view(n, 1, 1, -1),
                                  dt.view(n, 1, 1, -1), ws, False, True, grad)
                d = dh
                continue
            if item[0] == 'conv':
                _, layer, x, y = item
                d = df.view(y[lo:hi].shape) if d is None else d
                if record is not None:
                    record.append(d)
                last = layer is first_layer and item is tape[0]
                d = layer.bwd(x[lo:hi], y[lo:hi], d, ws, need_dx or not last, need_dw, grad)
            else:
                _, a, b, h, r1, r2, out = item
                d = df.view(out[lo:hi].shape) if d is None else d
                if record is not None:
                    dpre = G.act_bwd(out[lo:hi], d, 'lrelu')          # d (r2 + h)
                    record.append(dpre)
                    dr1 = b.bwd(r1[lo:hi], None, dpre, ws, True, need_dw, grad)
                    record.append(dr1)
                    dh = a.bwd(h[lo:hi], r1[lo:hi], dr1, ws, True, need_dw, grad)
                    d = G.axpy(1.0, dpre, dh)
                else:
                    # the join's lrelu slope is applied inside conv b's backward (its dy loaders multiply by act'(y) anyway:
                    # y = the join's output) and once more in the fused add at the end: two launches fewer per block
                    dr1 = b.bwd(r1[lo:hi], out[lo:hi], d, ws, True, need_dw, grad, act='lrelu')
                    dh = a.bwd(h[lo:hi], r1[lo:hi], dr1, ws, True, need_dw, grad)
                    d = G.act_bwd_add(out[lo:hi], d, 'lrelu', dh)     # dh += d * lrelu'(out)
        ws.join()
        return d.view(n, -1) if (need_dx and d is not None) else None

    def gp_grads(self, tapef, lo, hi, lambda_gp, loss_out, grad, ws=None):
        """Gradient penalty lambda * mean((||dD/dX_hat|| - 1)^2) over the rows [lo, hi) of a recorded
        forward (the X_hat rows) and its gradient w.r.t. the discriminator variables, into `grad`.

        With u_l the backward signal (u = d disc / d layer input) the penalty depends on the weights
        only through the linear maps of the backward chain u_l = A_l^T (s_l * u_{l+1}) (the lrelu slopes
        s_l are piecewise constant, the biases drop out).  Its adjoint is a forward pass
        v_{l+1} = s_l * (A_l v_l) started from v_0 = d penalty / d u_0, and
        d penalty / d A_l = (s_l * u_{l+1}) (x) v_l -- the ordinary weight gradient with the layer
        input replaced by v_l and the upstream gradient by the recorded u_{l+1}."""
        tape, f = tapef
        n = hi - lo
        ws = self.ws if ws is None else ws
        ones = torch.ones(n, 1, dtype=torch.float32, device=self.device)
        rec = []
        # entries no product below writes (biases: they drop out of the penalty; the latent head; fc_t) must read zero: ONE fill
        # of the flat buffer in front of everything instead of a fill per bias vector behind it (25 launches per step)
        grad.zero_()
        u0 = self.backward(tapef, lo, hi, ones, None, True, False, record=rec, ws=ws)
        v = G.grad_penalty(u0.contiguous(), lambda_gp, loss_out, need_v=True)
        # rec was filled from the head back to the input; walk it in forward order
        it = iter(reversed(rec))
        v = v.view((n,) + self.ae_shape)
        for item in tape:
            if item[0] == 'tcat':
                v = G.tile_concat_channels(v.contiguous(), None, self.t_channels)   # t does not depend on X_hat
                continue
            if item[0] == 'conv':
                _, layer, x, y = item
                dy = next(it)
                layer.bwd(v, y[lo:hi], dy, ws, False, True, grad, need_db=False)
                lin = layer.fwd(v, ws, act=None, bias=False)
                v = G.act_bwd(y[lo:hi], lin, layer.act) if layer.act else lin
            else:
                _, a, b, h, r1, r2, out = item
                dr1 = next(it)     # upstream of conv a (recorded after dpre, so it comes first in reverse)
                dpre = next(it)    # upstream of conv b
                a.bwd(v, r1[lo:hi], dr1, ws, False, True, grad, need_db=False)
                lin = a.fwd(v, ws, act=None, bias=False)
                v1 = G.act_bwd(r1[lo:hi], lin, 'lrelu')
                b.bwd(v1, None, dpre, ws, False, True, grad, need_db=False)
                v2 = b.fwd(v1, ws, act=None, bias=False)
                G.axpy(1.0, v, v2)                                   # v_r2 + v_h
                v = G.act_bwd(out[lo:hi], v2, 'lrelu')
        dy = next(it)
        self.disc_head.bwd(v.reshape(n, 1, 1, self.feat), None, dy, ws, False, True, grad, need_db=False)
        ws.join()
