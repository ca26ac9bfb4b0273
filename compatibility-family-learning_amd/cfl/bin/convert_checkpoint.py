"""python -m cfl.bin.convert_checkpoint -- exchange checkpoints with the reference.

The native checkpoint (`model-<step>.pt`) already stores every variable under its TensorFlow name and in the
reference layout, so conversion is a name-for-name copy:

    --to-npz   model-N.pt  out.npz        variables (+ Adam slots as <name>/Adam, <name>/Adam_1) as NumPy arrays
    --from-npz in.npz      model-N.pt     the reverse (Adam slots optional; global step from --step)
    --to-tf    model-N.pt  out_prefix     a TensorFlow checkpoint (tensor bundle: out_prefix.index + .data-00000-of-00001
                                          + the directory's `checkpoint` file)
    --from-tf  in_prefix   model-N.pt     read a checkpoint written by the reference's tf.train.Saver

The two TensorFlow directions are EXPERIMENTAL: they need no TensorFlow -- cfl/tf_bundle.py reads and writes the bundle format
itself (round 5) -- but neither the byte format nor the name mapping below has ever been read or written by a real TensorFlow
(none is installable in the build image).  The .npz directions are the supported exchange.
Names on the TensorFlow side follow what TF-1 creates for the reference's graph (cfl/models/dist.py:127-189,
cfl/models/cfl.py:1065-1096 -- the optimisers are built INSIDE the model's variable scope):
    variables        <scope>/...                          as stored here
    Adam slots       <scope>/<scope>/.../<var>/Adam, /Adam_1   (slot_creator nests the variable's full name under the CURRENT
                                                          scope: the well-known doubled prefix of TF-1 checkpoints)
    power variables  <scope>/beta1_power, <scope>/beta2_power; a graph with several AdamOptimizers numbers them in creation
                     order: [th_optim when the threshold has its own optimiser,] s_optim, post_g_optim, post_d_optim ->
                     beta1_power, beta1_power_1, ...
What a bundle written here does NOT contain: the ExponentialMovingAverage shadow variables of the reference's display
statistics (cfl/models/cfl.py:528; their names derive from TensorFlow's op numbering) -- so the reference's warm start
(`load_pre_weights`: assign_from_checkpoint_fn over the trainable variables, cfl/utils.py:478-490) and any
Saver(var_list=...) read it, a blanket `tf.train.Saver().restore` asks for those shadows too.  Not verified against a real
TensorFlow (none is installable in the build image); writer and reader are tested against each other and the format's
known answers (tests/test_tf_bundle.py).
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
    for n, (b1, b2) in (state.get('gan_powers') or {}).items():       # the two Adams of the MrCGAN post epochs
        out['gan_beta1_power_' + n], out['gan_beta2_power_' + n] = np.float32(b1), np.float32(b2)
    return out


def arrays_to_state(arrays, step=None, name=None):
    variables, m, v = {}, {}, {}
    for key in arrays:
        if key in ('beta1_power', 'beta2_power', 'global_step') or key.startswith('gan_beta'):
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
    state = {'variables': variables, 'adam_m': m, 'adam_v': v,
             'beta1_power': float(arrays['beta1_power']) if 'beta1_power' in arrays else 0.9,
             'beta2_power': float(arrays['beta2_power']) if 'beta2_power' in arrays else 0.999,
             'global_step': gs if step is None else int(step), 'name': name or ''}
    if 'gan_beta1_power_g' in arrays:
        state['gan_powers'] = {n: (float(arrays['gan_beta1_power_' + n]), float(arrays['gan_beta2_power_' + n]))
                               for n in ('g', 'd')}
    return state


def _scope_of(name):
    return name.split('/', 1)[0]


def to_tf_names(arrays, model_name='', own_threshold_optimiser=None):
    """flat exchange names (<var>, <var>/Adam, <var>/Adam_1, beta*_power, gan powers) -> TF-1 checkpoint names.
    own_threshold_optimiser: does the graph have the separate `th_optim` Adam on the threshold (CFL without --use-threshold,
    cfl/models/cfl.py:1076-1079)?  Stated by the checkpoint (state['graph']); None = checkpoints written before round 6:
    inferred from the run name (`_ut` in get_name())."""
    out = {}
    scope = None
    for key, val in arrays.items():
        if key in ('beta1_power', 'beta2_power', 'global_step') or key.startswith('gan_beta'):
            continue
        base = key
        for suffix in ADAM_SUFFIX[::-1]:
            if key.endswith(suffix):
                base = key[:-len(suffix)]
                break
        scope = scope or _scope_of(base)
        out[key if base == key else _scope_of(base) + '/' + key] = val
    scope = scope or 'CFL'
    # power accumulators in the optimisers' creation order
    if own_threshold_optimiser is None:
        own_threshold_optimiser = scope == 'CFL' and '_ut' not in ('_' + model_name + '_').replace('_reg', '_') and bool(model_name)
    chain = [('beta1_power', 'beta2_power')] * (2 if own_threshold_optimiser else 1)
    if 'gan_beta1_power_g' in arrays:
        chain += [('gan_beta1_power_g', 'gan_beta2_power_g'), ('gan_beta1_power_d', 'gan_beta2_power_d')]
    for i, (k1, k2) in enumerate(chain):
        sfx = '' if i == 0 else '_%d' % i
        out['%s/beta1_power%s' % (scope, sfx)] = np.float32(arrays.get(k1, 0.9))
        out['%s/beta2_power%s' % (scope, sfx)] = np.float32(arrays.get(k2, 0.999))
    # the step counter of the run (the reference has no global_step variable of its own -- the Saver's global_step only names
    # the file, cfl/utils.py:476-477 -- so this is an extra tensor a Saver(var_list=...) of the reference never asks for)
    if 'global_step' in arrays:
        out['global_step'] = np.int64(arrays['global_step'])
    return out


def from_tf_names(tensors):
    """TF-1 checkpoint names -> the flat exchange names; ExponentialMovingAverage shadows are dropped"""
    out = {}
    powers = {}
    for key, val in tensors.items():
        if key.endswith('/ExponentialMovingAverage'):
            continue
        tail = key.rsplit('/', 1)[-1]
        if tail.startswith('beta1_power') or tail.startswith('beta2_power'):
            idx = int(tail.split('_')[-1]) if tail.count('_') == 2 else 0
            powers.setdefault(idx, {})[tail[:11]] = np.float32(val)
            continue
        if key == 'global_step' or key.endswith('/global_step'):
            out['global_step'] = np.int64(val)
            continue
        name = key
        for suffix in ADAM_SUFFIX[::-1]:
            if key.endswith(suffix):
                scope = _scope_of(key)
                if key.startswith(scope + '/' + scope + '/'):      # the doubled prefix of slots created inside the scope
                    name = key[len(scope) + 1:]
                break
        out[name] = val
    if powers:
        order = sorted(powers)
        # the encoder's optimiser is the first one -- or the second when the threshold has an optimiser of its own, whose
        # accumulators carry the same values; the last two of a graph with four or three-plus-one are the GAN's (g, d)
        main = powers[order[0]]
        out['beta1_power'], out['beta2_power'] = main.get('beta1_power', 0.9), main.get('beta2_power', 0.999)
        if len(order) >= 3:
            g, d = powers[order[-2]], powers[order[-1]]
            out['gan_beta1_power_g'], out['gan_beta2_power_g'] = g.get('beta1_power', 0.5), g.get('beta2_power', 0.999)
            out['gan_beta1_power_d'], out['gan_beta2_power_d'] = d.get('beta1_power', 0.5), d.get('beta2_power', 0.999)
    return out


def to_tf(arrays, prefix, model_name='', own_threshold_optimiser=None):
    from .. import tf_bundle
    tf_bundle.write_bundle(prefix, to_tf_names(arrays, model_name, own_threshold_optimiser))


def from_tf(prefix):
    from .. import tf_bundle
    return from_tf_names(tf_bundle.read_bundle(prefix))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    g = ap.add_mutually_exclusive_group(required=True)
    for flag, what in (('--to-npz', 'native checkpoint -> NumPy archive under the TensorFlow variable names'),
                       ('--from-npz', 'NumPy archive -> native checkpoint'),
                       ('--to-tf', 'EXPERIMENTAL: native checkpoint -> TensorFlow tensor bundle (never read by a real TensorFlow)'),
                       ('--from-tf', 'EXPERIMENTAL: TensorFlow tensor bundle -> native checkpoint (never fed a real TF file)')):
        g.add_argument(flag, action='store_true', help=what)
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
        st = _load_pt(a.src)
        graph = st.get('graph') or {}
        to_tf(state_to_arrays(st), a.dst, st.get('name', ''), graph.get('own_threshold_optimiser'))
    else:
        import torch
        torch.save(Saver._plain(arrays_to_state(from_tf(a.src), a.step)), a.dst)
    return 0


if __name__ == '__main__':
    sys.exit(main())
