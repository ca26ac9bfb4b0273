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
        self.mul, self.add, self.lo, self.hi = float(mul), float(add), lo, hi
        self.input_shape = tuple(input_shape) if input_shape else None
        self.spec = spec or {}

    def __call__(self, x):
        x = np.asarray(x)
        y = x * x.dtype.type(self.mul) + x.dtype.type(self.add)
        if self.lo is not None or self.hi is not None:
            y = np.clip(y, self.lo, self.hi)
        if self.input_shape:
            y = y.reshape(-1, int(np.prod(self.input_shape)))
        return y

    def to_cfl_norm(self):
        from . import hipabi
        return hipabi.make_norm(self.mul, self.add, self.lo, self.hi)

    def inverse(self):
        return Unnormalizer(self.mul, self.add, self.input_shape)


class Unnormalizer(object):
    def __init__(self, mul, add, input_shape=None):
        self.mul, self.add, self.input_shape = float(mul), float(add), input_shape

    def __call__(self, y):
        y = np.asarray(y)
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


def _scalar(v, what):
    if v is None:
        return None
    if isinstance(v, (list, tuple)):
        if len(v) != 1:
            raise NotImplementedError(
                'per-channel %s (cfl/ops.py:84-106) belongs to the image path, which is '
                'outside the linear hot path' % what)
        return float(v[0])
    return float(v)


def normalizer_v2(input_shape, scale=None, mean=None, norm=None, clip_value_min=None,
                  clip_value_max=None):
    """cfl/ops.py:66-143, scalar mean / norm: ((x * scale) - mean) / norm, clip,
    reshape(-1, prod(input_shape))."""
    scale = 1.0 if scale is None else float(scale)
    mean = _scalar(mean, 'mean') or 0.0
    norm = _scalar(norm, 'norm') or 1.0
    return Normalizer(scale / norm, -mean / norm, clip_value_min, clip_value_max, input_shape,
                      spec=dict(scale=scale, mean=mean, norm=norm))


def normalize_v2(tensor, input_shape, scale=None, mean=None, norm=None, clip_value_min=None,
                 clip_value_max=None):
    return normalizer_v2(input_shape, scale, mean, norm, clip_value_min, clip_value_max)(tensor)


def unnormalizer_v2(input_shape, scale=None, mean=None, norm=None):
    """cfl/ops.py:146-195, scalar case: (y * norm + mean) / scale."""
    return normalizer_v2(None, scale, mean, norm).inverse() if input_shape is None else \
        Unnormalizer((1.0 if scale is None else float(scale)) / (_scalar(norm, 'norm') or 1.0),
                     -(_scalar(mean, 'mean') or 0.0) / (_scalar(norm, 'norm') or 1.0), input_shape)


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
