"""python -m cfl.bin.convert_checkpoint -- exchange checkpoints with the reference.

The native checkpoint (`model-<step>.pt`) already stores every variable under its TensorFlow name and in the
reference layout, so conversion is a name-for-name copy:

    --to-npz   model-N.pt  out.npz        variables (+ Adam slots as <name>/Adam, <name>/Adam_1) as NumPy arrays
    --from-npz in.npz      model-N.pt     the reverse (Adam slots optional; global step from --step)
    --to-tf    model-N.pt  out_prefix     a TensorFlow bundle readable by the reference's tf.train.Saver
    --from-tf  in_prefix   model-N.pt     read a reference checkpoint (tf.train.load_checkpoint)

The two TensorFlow directions need an importable `tensorflow` (1.x or 2.x with compat.v1); this image has none,
so they fail loudly here and are exercised only through the .npz path in the tests.
"""
import argparse
import sys

import numpy as np

from ..utils import Saver

ADAM_SUFFIX = ('/Adam', '/Adam_1')      # TF-1 slot names of tf.train.AdamOptimizer (m, v)


def _load_pt(path):
    from ..utils import load_checkpoint_file
    return load_checkpoint_file(path)


def state_to_arrays(state):
    out = {}
    for name, value in state['variables'].items():
        out[name] = np.asarray(value, np.float32)
    for key, suffix in (('adam_m', ADAM_SUFFIX[0]), ('adam_v', ADAM_SUFFIX[1])):
        for name, value in state.get(key, {}).items():
            out[name + suffix] = np.asarray(value, np.float32)
    out['beta1_power'] = np.float32(state.get('beta1_power', 0.9))
    out['beta2_power'] = np.float32(state.get('beta2_power', 0.999))
    out['global_step'] = np.int64(state.get('global_step', 0))
    return out


def arrays_to_state(arrays, step=None, name=None):
    variables, m, v = {}, {}, {}
    for key in arrays:
        if key in ('beta1_power', 'beta2_power', 'global_step'):
            continue
        if key.endswith(ADAM_SUFFIX[1]):
            v[key[:-len(ADAM_SUFFIX[1])]] = np.asarray(arrays[key], np.float32)
        elif key.endswith(ADAM_SUFFIX[0]):
            m[key[:-len(ADAM_SUFFIX[0])]] = np.asarray(arrays[key], np.float32)
        else:
            variables[key] = np.asarray(arrays[key], np.float32)
    for k, val in variables.items():          # a reference checkpoint saved without slots: start Adam fresh
        m.setdefault(k, np.zeros_like(val))
        v.setdefault(k, np.zeros_like(val))
    gs = int(arrays['global_step']) if 'global_step' in arrays else 0
    return {'variables': variables, 'adam_m': m, 'adam_v': v,
            'beta1_power': float(arrays['beta1_power']) if 'beta1_power' in arrays else 0.9,
            'beta2_power': float(arrays['beta2_power']) if 'beta2_power' in arrays else 0.999,
            'global_step': gs if step is None else int(step), 'name': name or ''}


def _tf():
    try:
        import tensorflow as tf
    except ImportError as e:       # pragma: no cover - no TensorFlow in the build image
        raise SystemExit('this direction needs TensorFlow (pip install tensorflow): %s' % e)
    return tf.compat.v1 if hasattr(tf, 'compat') and hasattr(tf.compat, 'v1') else tf


def to_tf(arrays, prefix):         # pragma: no cover
    tf = _tf()
    tf.disable_eager_execution() if hasattr(tf, 'disable_eager_execution') else None
    with tf.Graph().as_default(), tf.Session() as sess:
        vs = [tf.Variable(val, name=name) for name, val in arrays.items()]
        sess.run(tf.variables_initializer(vs))
        tf.train.Saver(vs).save(sess, prefix, write_meta_graph=False)


def from_tf(prefix):               # pragma: no cover
    tf = _tf()
    reader = tf.train.load_checkpoint(prefix)
    return {name: reader.get_tensor(name) for name in reader.get_variable_to_shape_map()}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    g = ap.add_mutually_exclusive_group(required=True)
    for flag in ('--to-npz', '--from-npz', '--to-tf', '--from-tf'):
        g.add_argument(flag, action='store_true')
    ap.add_argument('src')
    ap.add_argument('dst')
    ap.add_argument('--step', type=int, default=None)
    a = ap.parse_args(argv)
    if a.to_npz:
        np.savez(a.dst, **state_to_arrays(_load_pt(a.src)))
    elif a.from_npz:
        import torch
        with np.load(a.src) as z:
            torch.save(Saver._plain(arrays_to_state({k: z[k] for k in z.files}, a.step)), a.dst)
    elif a.to_tf:
        to_tf(state_to_arrays(_load_pt(a.src)), a.dst)
    else:
        import torch
        torch.save(Saver._plain(arrays_to_state(from_tf(a.src), a.step)), a.dst)
    return 0


if __name__ == '__main__':
    sys.exit(main())
