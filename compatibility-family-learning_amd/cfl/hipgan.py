"""ctypes binding of the MrCGAN part of libcfl_hip.so (include/cfl_hip.h: transposed conv,
element-wise / permutation / loss kernels).  Same rules as cfl/hipabi.py: tensors must be
contiguous fp32 on the GPU, no CPU fallback."""
import ctypes as C
import os

import torch

from . import hipabi as H
from .hipabi import CflConv, _check, _dev, _stream

EXPORTS = ('cfl_conv_transpose_workspace_bytes', 'cfl_conv2d_transpose_wn_fwd', 'cfl_conv2d_transpose_wn_bwd',
           'cfl_ew_act_fwd', 'cfl_ew_act_bwd', 'cfl_ew_act_bwd_add', 'cfl_ew_add_act', 'cfl_ew_axpy', 'cfl_ew_affine_clip', 'cfl_subpixel2x_fwd',
           'cfl_subpixel2x_bwd', 'cfl_concat_cols', 'cfl_gather_prototype', 'cfl_bce_logits',
           'cfl_rowdist_loss', 'cfl_perturb_workspace_bytes', 'cfl_perturb', 'cfl_grad_penalty', 'cfl_copy_cols',
           'cfl_tile_concat_channels', 'cfl_tile_concat_channels_bwd', 'cfl_auc_workspace_bytes', 'cfl_auc', 'cfl_image_transform',
           'cfl_ew_affine_clip_channels', 'cfl_conv_cache_bytes', 'cfl_conv2d_wn_fwd_cached', 'cfl_conv2d_wn_bwd_cached', 'cfl_conv2d_wn_fwd_fused', 'cfl_conv2d_wn_bwd_fused',
           'cfl_conv_bwd_takes_subpixel', 'cfl_conv_prepare_cached', 'cfl_conv_wgrad_slab_bytes', 'cfl_conv2d_wn_wgrad_slabs',
           'cfl_conv_wfinal_many', 'cfl_conv_prepare_cached_many')

EW = {None: 0, 'linear': 0, 'lrelu': 1, 'relu': 2, 'tanh': 3, 'sigmoid': 4}

_ready = False


def lib():
    global _ready
    L = H.lib()
    if _ready:
        return L
    vp, i64, i32, f32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t
    L.cfl_conv_transpose_workspace_bytes.argtypes = [C.POINTER(CflConv)]
    L.cfl_conv_transpose_workspace_bytes.restype = sz
    L.cfl_conv2d_transpose_wn_fwd.argtypes = [C.POINTER(CflConv)] + [vp] * 6 + [sz, vp]
    L.cfl_conv2d_transpose_wn_bwd.argtypes = [C.POINTER(CflConv)] + [vp] * 5 + [f32] + [vp] * 5 + [sz, vp]
    L.cfl_ew_act_fwd.argtypes = [vp, vp, i64, i32, vp]
    L.cfl_ew_act_bwd.argtypes = [vp, vp, vp, i64, i32, vp]
    L.cfl_ew_add_act.argtypes = [vp, vp, vp, i64, i32, vp]
    L.cfl_ew_act_bwd_add.argtypes = [vp, vp, vp, i64, i32, vp]
    L.cfl_ew_axpy.argtypes = [f32, vp, vp, i64, vp]
    L.cfl_ew_affine_clip.argtypes = [vp, vp, i64, C.POINTER(H.CflNorm), vp]
    L.cfl_subpixel2x_fwd.argtypes = [vp, vp, i64, i32, i32, i32, i32, vp]
    L.cfl_subpixel2x_bwd.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, vp]
    L.cfl_concat_cols.argtypes = [vp, i32, vp, i32, i64, vp, vp]
    L.cfl_copy_cols.argtypes = [vp, i32, i32, vp, i32, i32, i32, i64, vp]
    L.cfl_tile_concat_channels.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp]
    L.cfl_tile_concat_channels_bwd.argtypes = [vp, i32, i32, i64, i32, vp, vp, vp]
    L.cfl_gather_prototype.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.cfl_bce_logits.argtypes = [vp, i64, f32, f32, vp, vp, vp, i32, vp]
    L.cfl_rowdist_loss.argtypes = [vp, vp, i64, i32, i32, f32, f32, vp, vp, i32, vp]
    L.cfl_perturb_workspace_bytes.argtypes = []
    L.cfl_perturb_workspace_bytes.restype = sz
    L.cfl_perturb.argtypes = [vp, vp, i64, i64, f32, vp, vp, sz, vp]
    L.cfl_grad_penalty.argtypes = [vp, i64, i64, f32, vp, vp, vp, vp]
    L.cfl_ew_affine_clip_channels.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                              C.POINTER(H.CflNorm), vp]
    L.cfl_image_transform.argtypes = [vp, i64, i32, i32, i32, vp, i32, i32, vp, vp, i32, vp]
    L.cfl_auc_workspace_bytes.argtypes = [i64, i64]
    L.cfl_auc_workspace_bytes.restype = sz
    L.cfl_auc.argtypes = [vp, i64, vp, i64, vp, vp, sz, vp]
    L.cfl_conv_cache_bytes.argtypes = [C.POINTER(CflConv)]
    L.cfl_conv_cache_bytes.restype = sz
    L.cfl_conv2d_wn_fwd_cached.argtypes = [C.POINTER(CflConv)] + [vp] * 6 + [sz, vp, sz, C.POINTER(C.c_int32), vp]
    L.cfl_conv2d_wn_bwd_cached.argtypes = ([C.POINTER(CflConv)] + [vp] * 5 + [f32] + [vp] * 5 +
                                           [sz, vp, sz, C.POINTER(C.c_int32), vp])
    L.cfl_conv2d_wn_fwd_fused.argtypes = ([C.POINTER(CflConv)] + [vp] * 5 + [C.c_int32, vp, vp, sz, vp, sz,
                                                                            C.POINTER(C.c_int32), vp])
    L.cfl_conv2d_wn_bwd_fused.argtypes = ([C.POINTER(CflConv)] + [vp] * 5 + [C.c_int32, f32] + [vp] * 5 +
                                          [sz, vp, sz, C.POINTER(C.c_int32), vp])
    L.cfl_conv_bwd_takes_subpixel.argtypes = [C.POINTER(CflConv)]
    L.cfl_conv_prepare_cached.argtypes = [C.POINTER(CflConv), vp, vp, vp, sz, C.POINTER(C.c_int32), vp]
    L.cfl_conv_prepare_cached_many.argtypes = [C.c_int32, C.POINTER(CflConv), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                               C.POINTER(sz), C.POINTER(C.POINTER(C.c_int32)), vp]
    L.cfl_conv_wgrad_slab_bytes.argtypes = [C.POINTER(CflConv)]
    L.cfl_conv_wgrad_slab_bytes.restype = sz
    L.cfl_conv2d_wn_wgrad_slabs.argtypes = [C.POINTER(CflConv), vp, vp, vp, C.c_int32, vp, sz, vp]
    L.cfl_conv_wfinal_many.argtypes = [C.c_int32, C.POINTER(CflConv)] + [C.POINTER(vp)] * 4 + [f32] + [C.POINTER(vp)] * 3 + [vp]
    for n in EXPORTS:
        if n not in ('cfl_conv_transpose_workspace_bytes', 'cfl_perturb_workspace_bytes', 'cfl_auc_workspace_bytes',
                     'cfl_conv_cache_bytes', 'cfl_conv_wgrad_slab_bytes'):
            getattr(L, n).restype = C.c_int
    _ready = True
    return L


def _opt(t):
    return _dev(t) if t is not None else None


# ---- layers writing into caller-provided buffers (views of the flat parameter / gradient arrays) ----
def conv_ws_bytes(conv, transposed=False):
    L = lib()
    n = (L.cfl_conv_transpose_workspace_bytes if transposed else L.cfl_conv_workspace_bytes)(C.byref(conv))
    if n == 0:
        raise H.CflHipError('conv workspace: ' + L.cfl_last_error().decode())
    return n


class ConvCache(object):
    """Per-layer cache of what depends on the weights only (weight-norm scale, prepared filter planes of the direct 3x3
    kernel): a device buffer + the host validity bits of include/cfl_hip.h (CFL_CONV_CACHE_*).  `invalidate()` whenever
    V or g change.  CFL_CONV_CACHE=0 in the environment disables caching (every call recomputes, as the plain ABI)."""
    enabled = os.environ.get('CFL_CONV_CACHE', '1') not in ('0', '')

    def __init__(self):
        self.buf = None
        self.flags = C.c_int32(0)

    def invalidate(self):
        self.flags.value = 0

    def ensure(self, conv, device):
        L = lib()
        n = L.cfl_conv_cache_bytes(C.byref(conv))
        if self.buf is None or self.buf.numel() * 4 < n:
            self.buf = torch.empty((n + 3) // 4, dtype=torch.float32, device=device)
            self.flags.value = 0
        return self


# Analytic flop count of the convolution products issued (bench.py's MrCGAN roofline figure): set `flop_counter = [0]`,
# run a step, read flop_counter[0].  One product (forward, input gradient or weight gradient) of a layer is
# 2 * B * OH * OW * KH * KW * Ci * Co flops (transposed: per INPUT pixel).  None = not counting.
flop_counter = None


def _count_flops(conv, transposed, products):
    if flop_counter is not None and products:
        oh, ow = (conv.H, conv.W) if transposed else (-(-conv.H // conv.stride), -(-conv.W // conv.stride))
        flop_counter[0] += products * 2 * conv.B * oh * ow * conv.KH * conv.KW * conv.Ci * conv.Co


def conv_fwd(conv, x, V, g, b, y, ws, transposed=False, cache=None, residual=None, subpixel=False):
    """residual / subpixel: store epilogues of cfl_conv2d_wn_fwd_fused (y = act(conv + b + residual); y stored 2x
    sub-pixel shuffled [B, 2OH, 2OW, Co/4]); plain convolutions only."""
    L = lib()
    _count_flops(conv, transposed, 1)
    if residual is not None or subpixel:
        if transposed:
            raise H.CflHipError('residual / sub-pixel epilogues exist for the plain convolution only')
        use = cache is not None and ConvCache.enabled
        if use:
            cache.ensure(conv, x.device)
        _check(L.cfl_conv2d_wn_fwd_fused(C.byref(conv), _dev(x), _dev(V), _opt(g), _opt(b), _opt(residual), int(bool(subpixel)),
                                         _dev(y), ws.data_ptr(), ws.numel() * 4, cache.buf.data_ptr() if use else None,
                                         cache.buf.numel() * 4 if use else 0, C.byref(cache.flags) if use else None, _stream()))
        return y
    if cache is not None and not transposed and ConvCache.enabled:
        cache.ensure(conv, x.device)
        _check(L.cfl_conv2d_wn_fwd_cached(C.byref(conv), _dev(x), _dev(V), _opt(g), _opt(b), _dev(y), ws.data_ptr(),
                                          ws.numel() * 4, cache.buf.data_ptr(), cache.buf.numel() * 4,
                                          C.byref(cache.flags), _stream()))
        return y
    fn = L.cfl_conv2d_transpose_wn_fwd if transposed else L.cfl_conv2d_wn_fwd
    _check(fn(C.byref(conv), _dev(x), _dev(V), _opt(g), _opt(b), _dev(y), ws.data_ptr(), ws.numel() * 4, _stream()))
    return y


def conv_prepare(conv, V, g, cache):
    """rebuild what is missing in a layer's cache (weight-norm scale, filter planes) without running the layer"""
    if cache is None or not ConvCache.enabled:
        return
    cache.ensure(conv, V.device)
    _check(lib().cfl_conv_prepare_cached(C.byref(conv), _dev(V), _opt(g), cache.buf.data_ptr(), cache.buf.numel() * 4,
                                         C.byref(cache.flags), _stream()))


def conv_wgrad_slab_bytes(conv):
    n = lib().cfl_conv_wgrad_slab_bytes(C.byref(conv))
    if n == 0:
        raise H.CflHipError('cfl_conv_wgrad_slab_bytes: ' + lib().cfl_last_error().decode())
    return n


def conv_wgrad_slabs(conv, x, y, dy, slab, dy_subpixel=False):
    """ONLY the weight-gradient contraction of a layer, as split-K slabs in `slab` (cfl_conv2d_wn_wgrad_slabs); the slabs are
    finished later, many layers at once, by conv_wfinal_many."""
    _count_flops(conv, False, 1)
    _check(lib().cfl_conv2d_wn_wgrad_slabs(C.byref(conv), _dev(x), _opt(y), _dev(dy), int(bool(dy_subpixel)), slab.data_ptr(),
                                           slab.numel() * 4, _stream()))


def conv_wfinal_many(jobs, reg_const=0.0):
    """jobs: [(conv, slab, V, g, cache, dV, dg, db)] -- the layers of one backward chain, finished by one slab-sum launch and one
    finalisation launch (cfl_conv_wfinal_many)"""
    n = len(jobs)
    if not n:
        return
    convs = (CflConv * n)(*[j[0] for j in jobs])
    arr = lambda k, f: (C.c_void_p * n)(*[f(j[k]) for j in jobs])
    ptr = lambda t: t.data_ptr() if t is not None else None
    _check(lib().cfl_conv_wfinal_many(n, convs, arr(1, ptr), arr(2, ptr), arr(3, ptr), arr(4, lambda c: c.buf.data_ptr()),
                                      float(reg_const), arr(5, ptr), arr(6, ptr), arr(7, ptr), _stream()))


def conv_prepare_many(layers):
    """layers: [(conv, V, g, cache)] -- rebuild what is missing in the caches of ALL of them with one launch for the weight-norm
    scales and one for the filter planes (cfl_conv_prepare_cached_many)"""
    if not ConvCache.enabled:
        return
    layers = [l for l in layers if l[3] is not None]
    n = len(layers)
    if not n:
        return
    for conv, V, g, cache in layers:
        cache.ensure(conv, V.device)
    convs = (CflConv * n)(*[l[0] for l in layers])
    vps = lambda f: (C.c_void_p * n)(*[f(l) for l in layers])
    flags = (C.POINTER(C.c_int32) * n)(*[C.pointer(l[3].flags) for l in layers])
    sizes = (C.c_size_t * n)(*[l[3].buf.numel() * 4 for l in layers])
    _check(lib().cfl_conv_prepare_cached_many(n, convs, vps(lambda l: l[1].data_ptr()), vps(lambda l: l[2].data_ptr() if l[2] is not None else None),
                                              vps(lambda l: l[3].buf.data_ptr()), sizes, flags, _stream()))


def conv_bwd_takes_subpixel(conv):
    """can conv_bwd(..., dy_subpixel=True) read dy / y in the 2x sub-pixel shuffled layout for this shape?"""
    return lib().cfl_conv_bwd_takes_subpixel(C.byref(conv)) == 1


def conv_bwd(conv, x, V, g, y, dy, ws, dx=None, dV=None, dg=None, db=None, reg_const=0.0, transposed=False, cache=None,
             dy_subpixel=False):
    """dy_subpixel: dy (and y) are [B, 2OH, 2OW, Co/4], the layout conv_fwd(subpixel=True) stores -- the un-shuffle of the
    gradient (subpixel_bwd) folded into the dy loaders of the halo-tile kernels (only where conv_bwd_takes_subpixel())."""
    L = lib()
    _count_flops(conv, transposed, int(dx is not None) + int(dV is not None))
    if dy_subpixel:
        if transposed:
            raise CflHipError('dy_subpixel: not for transposed convolutions')
        has_cache = cache is not None and ConvCache.enabled
        if has_cache:
            cache.ensure(conv, dy.device)
        _check(L.cfl_conv2d_wn_bwd_fused(C.byref(conv), _opt(x), _dev(V), _opt(g), _opt(y), _dev(dy), 1, float(reg_const),
                                         _opt(dx), _opt(dV), _opt(dg), _opt(db), ws.data_ptr(), ws.numel() * 4,
                                         cache.buf.data_ptr() if has_cache else None, cache.buf.numel() * 4 if has_cache else 0,
                                         C.byref(cache.flags) if has_cache else None, _stream()))
        return dx
    if cache is not None and not transposed and ConvCache.enabled:
        cache.ensure(conv, dy.device)
        _check(L.cfl_conv2d_wn_bwd_cached(C.byref(conv), _opt(x), _dev(V), _opt(g), _opt(y), _dev(dy), float(reg_const),
                                          _opt(dx), _opt(dV), _opt(dg), _opt(db), ws.data_ptr(), ws.numel() * 4,
                                          cache.buf.data_ptr(), cache.buf.numel() * 4, C.byref(cache.flags), _stream()))
        return dx
    fn = L.cfl_conv2d_transpose_wn_bwd if transposed else L.cfl_conv2d_wn_bwd
    _check(fn(C.byref(conv), _opt(x), _dev(V), _opt(g), _opt(y), _dev(dy), float(reg_const), _opt(dx), _opt(dV),
              _opt(dg), _opt(db), ws.data_ptr(), ws.numel() * 4, _stream()))
    return dx


# ---- element-wise glue ----------------------------------------------------------------------------
def act_fwd(x, act, out=None):
    out = torch.empty_like(x) if out is None else out
    _check(lib().cfl_ew_act_fwd(_dev(x), _dev(out), x.numel(), EW[act], _stream()))
    return out


def act_bwd(y, dy, act, out=None):
    out = torch.empty_like(dy) if out is None else out
    _check(lib().cfl_ew_act_bwd(_dev(y), _dev(dy), _dev(out), dy.numel(), EW[act], _stream()))
    return out


def act_bwd_add(y, dy, act, acc):
    """acc += dy * act'(y)"""
    _check(lib().cfl_ew_act_bwd_add(_dev(y), _dev(dy), _dev(acc), dy.numel(), EW[act], _stream()))
    return acc


def add_act(a, b, act, out=None):
    out = torch.empty_like(a) if out is None else out
    _check(lib().cfl_ew_add_act(_dev(a), _dev(b), _dev(out), a.numel(), EW[act], _stream()))
    return out


def axpy(alpha, x, y):
    _check(lib().cfl_ew_axpy(float(alpha), _dev(x), _dev(y), x.numel(), _stream()))
    return y


def affine_clip(x, norm, out=None):
    """norm: hipabi.CflNorm (Normalizer.to_cfl_norm())."""
    out = torch.empty_like(x) if out is None else out
    _check(lib().cfl_ew_affine_clip(_dev(x), _dev(out), x.numel(), C.byref(norm), _stream()))
    return out


def affine_clip_channels(x, mul, add, clip, out=None):
    """per-channel y = clip(x * mul[c] + add[c]) for NHWC rows (c = element index mod len(mul) <= 4)."""
    out = torch.empty_like(x) if out is None else out
    n = len(mul)
    m = (C.c_float * n)(*[float(v) for v in mul])
    a = (C.c_float * n)(*[float(v) for v in add])
    _check(lib().cfl_ew_affine_clip_channels(_dev(x), _dev(out), x.numel(), n, m, a, C.byref(clip), _stream()))
    return out


def subpixel_fwd(x, act=None):
    B, Hh, W, Cc = x.shape
    y = torch.empty(B, 2 * Hh, 2 * W, Cc // 4, dtype=torch.float32, device=x.device)
    _check(lib().cfl_subpixel2x_fwd(_dev(x), _dev(y), B, Hh, W, Cc, EW[act], _stream()))
    return y


def subpixel_bwd(y, dy, act=None):
    B, H2, W2, Cq = dy.shape
    dx = torch.empty(B, H2 // 2, W2 // 2, Cq * 4, dtype=torch.float32, device=dy.device)
    _check(lib().cfl_subpixel2x_bwd(_opt(y) if EW[act] else None, _dev(dy), _dev(dx), B, H2 // 2, W2 // 2, Cq * 4,
                                    EW[act], _stream()))
    return dx


def concat_cols(a, b, out=None):
    if out is None:
        out = torch.empty(a.shape[0], a.shape[1] + b.shape[1], dtype=torch.float32, device=a.device)
    _check(lib().cfl_concat_cols(_dev(a), a.shape[1], _dev(b), b.shape[1], a.shape[0], _dev(out), _stream()))
    return out


def copy_cols(src, src_off, dst, dst_off, ncols):
    """dst[:, dst_off:dst_off+ncols] = src[:, src_off:src_off+ncols] for row-major 2-D tensors."""
    _check(lib().cfl_copy_cols(_dev(src), src.shape[1], src_off, _dev(dst), dst.shape[1], dst_off, ncols,
                               src.shape[0], _stream()))
    return dst


def tile_concat_channels(h, t, C2=None):
    """h [N,H,W,C1], t [N,C2] (or None: C2 zero channels) -> [N,H,W,C1+C2]."""
    N, Hh, W, C1 = h.shape
    C2 = t.shape[1] if t is not None else C2
    out = torch.empty(N, Hh, W, C1 + C2, dtype=torch.float32, device=h.device)
    _check(lib().cfl_tile_concat_channels(_dev(h), C1, _opt(t), C2, N, Hh * W, _dev(out), _stream()))
    return out


def tile_concat_channels_bwd(d, C1, need_dh=True, need_dt=True):
    N, Hh, W, C = d.shape
    C2 = C - C1
    dh = torch.empty(N, Hh, W, C1, dtype=torch.float32, device=d.device) if need_dh else None
    dt = torch.empty(N, C2, dtype=torch.float32, device=d.device) if need_dt else None
    _check(lib().cfl_tile_concat_channels_bwd(_dev(d), C1, C2, N, Hh * W, _opt(dh), _opt(dt), _stream()))
    return dh, dt


def gather_prototype(P, c):
    """P [B,K,L] fp32, c [B] int32 -> [B,L]."""
    B, K, Ld = P.shape
    out = torch.empty(B, Ld, dtype=torch.float32, device=P.device)
    _check(lib().cfl_gather_prototype(_dev(P), _dev(c, torch.int32), B, K, Ld, _dev(out), _stream()))
    return out


def bce_logits(logits, label, weight, loss, frac_pos=None, dlogits=None, accumulate=False):
    """loss / frac_pos: 1-element device views."""
    _check(lib().cfl_bce_logits(_dev(logits), logits.numel(), float(label), float(weight), _opt(loss),
                                _opt(frac_pos), _opt(dlogits), int(bool(accumulate)), _stream()))


def rowdist_loss(a, b, mode, margin, weight, loss, da=None, accumulate=False):
    B, Ld = a.shape
    _check(lib().cfl_rowdist_loss(_dev(a), _dev(b), B, Ld, int(mode), float(margin or 0.0), float(weight),
                                  _opt(loss), _opt(da), int(bool(accumulate)), _stream()))


def perturb(x, eps, lambda_dra, out=None):
    B, N = x.shape
    out = torch.empty_like(x) if out is None else out
    ws = torch.empty(lib().cfl_perturb_workspace_bytes() // 4, dtype=torch.float32, device=x.device)
    _check(lib().cfl_perturb(_dev(x), _dev(eps), B, N, float(lambda_dra), _dev(out), ws.data_ptr(), ws.numel() * 4,
                             _stream()))
    return out


def grad_penalty(u, lambda_gp, loss, need_v=True):
    B, N = u.shape
    v = torch.empty_like(u) if need_v else None
    rowloss = torch.empty(B, dtype=torch.float32, device=u.device)
    _check(lib().cfl_grad_penalty(_dev(u), B, N, float(lambda_gp), _dev(loss), _opt(v), _dev(rowloss), _stream()))
    return v


def auc(scores_pos, scores_neg):
    """(auc, accuracy) of device score vectors (include/cfl_hip.h cfl_auc); one host read-back of 16 bytes."""
    n = lib().cfl_auc_workspace_bytes(scores_pos.numel(), scores_neg.numel())
    if n == 0:
        raise H.CflHipError('cfl_auc: bad sizes')
    ws = torch.empty((n + 7) // 8, dtype=torch.float64, device=scores_pos.device)
    out = torch.empty(2, dtype=torch.float64, device=scores_pos.device)
    _check(lib().cfl_auc(_dev(scores_pos), scores_pos.numel(), _dev(scores_neg), scores_neg.numel(),
                         out.data_ptr(), ws.data_ptr(), ws.numel() * 8, _stream()))
    a, acc = out.cpu().tolist()
    return a, acc


def image_transform(x, out_hw, offsets=None, flip=None, resize=False):
    """x [B,H,W,C] -> [B,h,w,C]: crop / pad window (offsets int32 [B,2] or central) or bilinear resize, then
    optional per-sample left-right flip (flip int32 [B])."""
    B, Hh, W, Cc = x.shape
    y = torch.empty(B, out_hw[0], out_hw[1], Cc, dtype=torch.float32, device=x.device)
    _check(lib().cfl_image_transform(_dev(x), B, Hh, W, Cc, _dev(y), out_hw[0], out_hw[1],
                                     _dev(offsets, torch.int32) if offsets is not None else None,
                                     _dev(flip, torch.int32) if flip is not None else None, int(bool(resize)),
                                     _stream()))
    return y
