"""Input normalisers of the cfl hot path (host-side descriptors).

The reference builds TensorFlow graph closures (cfl/ops.py:66-225, 302-349); here
the same factory names return small callable descriptors: calling one on a NumPy
array applies the arithmetic on the host (used by tests / tools), while
``.to_cfl_norm()`` hands the same affine+clip map to the HIP kernels, which apply
it to every input vector as it is loaded (include/cfl_hip.h CflNorm) -- there is no
separate normalisation pass over the batch on the device.
"""
import numpy as np


def lrelu(x, leak=0.2):
    """cfl/ops.py:10-12: relu(x) - leak * relu(-x)."""
    x = np.asarray(x)
    return np.maximum(x, 0) - leak * np.maximum(-x, 0)


class Normalizer(object):
    """x_hat = clip(x * mul + add, lo, hi), optionally reshaped to (-1, prod(shape))."""

    def __init__(self, mul=1.0, add=0.0, lo=None, hi=None, input_shape=None, spec=None):
        # per-channel (cfl/ops.py:84-106): mul / add are sequences over the last (channel) axis of input_shape
        self.per_channel = isinstance(mul, (list, tuple, np.ndarray))
        if self.per_channel:
            self.mul = [float(v) for v in mul]
            self.add = [float(v) for v in add]
        else:
            self.mul, self.add = float(mul), float(add)
        self.lo, self.hi = lo, hi
        self.input_shape = tuple(input_shape) if input_shape else None
        self.spec = spec or {}

    def __call__(self, x):
        x = np.asarray(x)
        if self.per_channel:
            c = len(self.mul)
            y = (x.reshape(-1, c) * np.asarray(self.mul, x.dtype) + np.asarray(self.add, x.dtype)).reshape(x.shape)
        else:
            y = x * x.dtype.type(self.mul) + x.dtype.type(self.add)
        if self.lo is not None or self.hi is not None:
            y = np.clip(y, self.lo, self.hi)
        if self.input_shape:
            y = y.reshape(-1, int(np.prod(self.input_shape)))
        return y

    def to_cfl_norm(self):
        from . import hipabi
        if self.per_channel:
            raise ValueError('a per-channel normaliser cannot be folded into the pair kernels; use apply()')
        return hipabi.make_norm(self.mul, self.add, self.lo, self.hi)

    def apply(self, t):
        """The normaliser as an explicit pass over a device tensor (images of the conv stacks, GAN inputs)."""
        from . import hipabi, hipgan
        if self.per_channel:
            return hipgan.affine_clip_channels(t.contiguous(), self.mul, self.add, hipabi.make_norm(1.0, 0.0, self.lo, self.hi))
        if self.mul == 1.0 and self.add == 0.0 and self.lo is None and self.hi is None:
            return t.contiguous()
        return hipgan.affine_clip(t.contiguous(), self.to_cfl_norm())

    def inverse(self):
        return Unnormalizer(self.mul, self.add, self.input_shape)


class Unnormalizer(object):
    def __init__(self, mul, add, input_shape=None):
        self.per_channel = isinstance(mul, (list, tuple, np.ndarray))
        self.mul = [float(v) for v in mul] if self.per_channel else float(mul)
        self.add = [float(v) for v in add] if self.per_channel else float(add)
        self.input_shape = input_shape

    def __call__(self, y):
        y = np.asarray(y)
        if self.per_channel:
            c = len(self.mul)
            x = ((y.reshape(-1, c) - np.asarray(self.add, y.dtype)) / np.asarray(self.mul, y.dtype)).reshape(y.shape)
        else:
            x = (y - y.dtype.type(self.add)) / y.dtype.type(self.mul)
        if self.input_shape:
            x = x.reshape((-1,) + tuple(self.input_shape))
        return x


def normalize(tensor, scale, shift, clip_value_min=None, clip_value_max=None):
    """cfl/ops.py:198-202: tensor / scale + shift, then clip."""
    return normalizer(scale, shift, clip_value_min, clip_value_max)(tensor)


def normalizer(scale, shift, clip_value_min=None, clip_value_max=None):
    """cfl/ops.py:205-214 (the Monomer path: scale = normalize_value, shift = 0)."""
    return Normalizer(1.0 / scale, shift, clip_value_min, clip_value_max,
                      spec=dict(scale=scale, shift=shift))


def unnormalize(tensor, scale, shift):
    return unnormalizer(scale, shift)(tensor)


def unnormalizer(scale, shift):
    """cfl/ops.py:217-225: (tensor - shift) * scale."""
    return Unnormalizer(1.0 / scale, shift)


def _channels(v, what, input_shape):
    """None, a scalar, a 1-list (scalar) or one value per channel of input_shape (cfl/ops.py:84-106)."""
    if v is None:
        return None
    if isinstance(v, (list, tuple, np.ndarray)):
        if len(v) == 1:
            return float(v[0])
        if input_shape is None or len(v) != tuple(input_shape)[-1]:
            raise ValueError('%s has %d values but the data has %r channels' % (what, len(v), input_shape))
        if len(v) > 4:
            raise NotImplementedError('per-channel %s for more than 4 channels' % what)
        return [float(t) for t in v]
    return float(v)


def normalizer_v2(input_shape, scale=None, mean=None, norm=None, clip_value_min=None,
                  clip_value_max=None):
    """cfl/ops.py:66-143: ((x * scale) - mean) / norm, clip, reshape(-1, prod(input_shape)); mean / norm scalar or
    one value per channel."""
    scale = 1.0 if scale is None else float(scale)
    mean = _channels(mean, 'mean', input_shape)
    norm = _channels(norm, 'norm', input_shape)
    if isinstance(mean, list) or isinstance(norm, list):
        c = tuple(input_shape)[-1]
        mean_c = mean if isinstance(mean, list) else [mean or 0.0] * c
        norm_c = norm if isinstance(norm, list) else [norm or 1.0] * c
        return Normalizer([scale / n for n in norm_c], [-m / n for m, n in zip(mean_c, norm_c)], clip_value_min,
                          clip_value_max, input_shape, spec=dict(scale=scale, mean=mean_c, norm=norm_c))
    mean, norm = mean or 0.0, norm or 1.0
    return Normalizer(scale / norm, -mean / norm, clip_value_min, clip_value_max, input_shape,
                      spec=dict(scale=scale, mean=mean, norm=norm))


def normalize_v2(tensor, input_shape, scale=None, mean=None, norm=None, clip_value_min=None,
                 clip_value_max=None):
    return normalizer_v2(input_shape, scale, mean, norm, clip_value_min, clip_value_max)(tensor)


def unnormalizer_v2(input_shape, scale=None, mean=None, norm=None):
    """cfl/ops.py:146-195, scalar case: (y * norm + mean) / scale."""
    n = normalizer_v2(input_shape, scale, mean, norm)
    return Unnormalizer(n.mul, n.add, input_shape)


CLIP_VALUES = {'sigmoid': (0., 1.), 'tanh': (-1., 1.), 'relu': (0., None), 'linear': (None, None)}


def dist_normalizer(input_shape, ae_shape, data_scale, data_mean, data_norm, latent_norm, data_type):
    """cfl/ops.py:302-349: (data_normalizer, data_unnormalizer, ae_normalizer,
    ae_unnormalizer, latent_normalizer) with the clip range of ``data_type``."""
    lo, hi = CLIP_VALUES[data_type]
    data_n = normalizer_v2(input_shape, data_scale, data_mean, data_norm, lo, hi)
    latent_n = normalizer_v2(None, None, None, latent_norm) if latent_norm else None
    data_u = unnormalizer_v2(input_shape, data_scale, data_mean, data_norm)
    if ae_shape is not None and tuple(ae_shape) != tuple(input_shape):
        ae_n = normalizer_v2(ae_shape, data_scale, data_mean, data_norm, lo, hi)
        ae_u = unnormalizer_v2(ae_shape, data_scale, data_mean, data_norm)
    else:
        ae_n, ae_u = data_n, data_u
    return data_n, data_u, ae_n, ae_u, latent_n


# ---------------------------------------------------------------------------
# input transformers (cfl/ops.py:28-63, 262-299): crop / random crop / resize / random mirror
# ---------------------------------------------------------------------------
class ImageTransform(object):
    """source_shape -> out_shape on flattened NHWC rows.  kind: 'crop' (central window of
    resize_image_with_crop_or_pad), 'random_crop' (tf.random_crop, one offset per sample), 'resize'
    (tf.image.resize_images, bilinear, align_corners=False), 'reshape' (shapes equal); `mirror` adds the
    per-sample tf.image.random_flip_left_right.  `apply` runs on the GPU (cfl_image_transform); calling the
    object on a NumPy array applies the deterministic part on the host (tests)."""

    def __init__(self, source_shape, out_shape, kind, mirror=False):
        self.source_shape = tuple(source_shape) if len(source_shape) == 3 else tuple(source_shape) + (1,)
        self.out_shape = tuple(out_shape) if len(out_shape) == 3 else tuple(out_shape) + (1,)
        self.kind, self.mirror = kind, bool(mirror)
        self.out_size = int(np.prod(self.out_shape))

    @property
    def is_identity(self):
        return self.kind == 'reshape' and not self.mirror

    def draw(self, n, rng):
        """(offsets [n,2] or None, flips [n] or None) from a NumPy RandomState."""
        off = flip = None
        if self.kind == 'random_crop':
            off = np.stack([rng.randint(0, self.source_shape[0] - self.out_shape[0] + 1, size=n),
                            rng.randint(0, self.source_shape[1] - self.out_shape[1] + 1, size=n)], 1).astype(np.int32)
        if self.mirror:
            flip = (rng.rand(n) < 0.5).astype(np.int32)
        return off, flip

    def apply(self, t, rng):
        """t: device tensor [n, prod(source_shape)] -> [n, prod(out_shape)]."""
        import torch
        from . import hipgan
        n = t.shape[0]
        x = t.contiguous().view((n,) + self.source_shape)
        off, flip = self.draw(n, rng)
        dev = lambda a: torch.from_numpy(a).to(t.device) if a is not None else None
        y = hipgan.image_transform(x, self.out_shape[:2], dev(off), dev(flip), resize=self.kind == 'resize')
        return y.view(n, self.out_size)

    def __call__(self, x, offsets=None, flips=None):
        x = np.asarray(x, np.float32).reshape((-1,) + self.source_shape)
        n, (H, W, C), (h, w, _) = x.shape[0], self.source_shape, self.out_shape
        y = np.zeros((n, h, w, C), np.float32)
        for b in range(n):
            if self.kind == 'resize':
                fy, fx = np.arange(h) * (np.float32(H) / np.float32(h)), np.arange(w) * (np.float32(W) / np.float32(w))
                y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
                y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
                ly, lx = (fy - y0).astype(np.float32)[:, None, None], (fx - x0).astype(np.float32)[None, :, None]
                top = x[b][y0][:, x0] + (x[b][y0][:, x1] - x[b][y0][:, x0]) * lx
                bot = x[b][y1][:, x0] + (x[b][y1][:, x1] - x[b][y1][:, x0]) * lx
                img = top + (bot - top) * ly
            else:
                oy = offsets[b][0] if offsets is not None else ((H - h) // 2 if H >= h else -((h - H) // 2))
                ox = offsets[b][1] if offsets is not None else ((W - w) // 2 if W >= w else -((w - W) // 2))
                img = np.zeros((h, w, C), np.float32)
                ys, xs = np.arange(h) + oy, np.arange(w) + ox
                vy, vx = (ys >= 0) & (ys < H), (xs >= 0) & (xs < W)
                img[np.ix_(vy, vx)] = x[b][np.ix_(ys[vy], xs[vx])]
            if flips is not None and flips[b]:
                img = img[:, ::-1]
            y[b] = img
        return y.reshape(n, -1)


def dist_transformer(source_shape, input_shape, data_random_crop, data_mirror):
    """(train_transformer, val_transformer) of cfl/ops.py:262-289; None stands for the plain reshape."""
    if source_shape is not None and tuple(input_shape) != tuple(source_shape):
        if source_shape[0] > input_shape[0]:
            train = ImageTransform(source_shape, input_shape, 'random_crop' if data_random_crop else 'crop', data_mirror)
            val = ImageTransform(source_shape, input_shape, 'crop')
        else:
            train = ImageTransform(source_shape, input_shape, 'resize', data_mirror)
            val = ImageTransform(source_shape, input_shape, 'resize')
        return train, val
    if data_mirror:
        return ImageTransform(input_shape, input_shape, 'reshape', True), None
    return None, None


def dist_ae_transformer(input_shape, ae_shape):
    """cfl/ops.py:292-299: resize to the generator / discriminator resolution when it differs."""
    if ae_shape is not None and tuple(input_shape) != tuple(ae_shape):
        return ImageTransform(input_shape, ae_shape, 'resize')
    return None
