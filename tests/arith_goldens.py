"""Access to the arithmetic goldens (tests/golden/arith_goldens.npz) captured from the reference's own
model classes (tests/golden/make_arith_goldens.py) + the maps between the reference's TensorFlow
variable names and the oracle's parameter dictionaries."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import arith_recipe as R  # noqa: E402

_npz = None
_meta = None


def npz():
    global _npz
    if _npz is None:
        _npz = np.load(os.path.join(HERE, 'golden', 'arith_goldens.npz'))
    return _npz


def meta(case_name):
    global _meta
    if _meta is None:
        with open(os.path.join(HERE, 'golden', 'arith_goldens_meta.json')) as f:
            _meta = {c['name']: c for c in json.load(f)['cases']}
    return _meta[case_name]


def has(case, step, key):
    k = '%s/step%d/%s' % (case['name'], step, key)
    return k in npz().files or k + '#digest' in npz().files


def check(case, step, key, arr, rtol=1e-9, atol=1e-11):
    """`arr` against the stored golden under key (whole array, or its seeded digest for large tensors)."""
    full = '%s/step%d/%s' % (case['name'], step, key)
    for k, mine in R.digest(full, np.asarray(arr, dtype=np.float64)).items():
        ref = npz()[k]
        assert mine.shape == ref.shape, (k, mine.shape, ref.shape)
        scale = max(float(np.abs(ref).max()), 1e-30) if ref.size else 1.0
        err = float(np.abs(mine - ref).max()) if ref.size else 0.0
        assert err <= atol + rtol * scale, '%s: max |diff| %.3e (scale %.3e)' % (k, err, scale)


def expected(case, step, key):
    return npz()['%s/step%d/%s' % (case['name'], step, key)]


def initial_variables(case):
    """{TF variable name: float64 array} exactly as the generator set them"""
    return {n: np.asarray(R.init_value(case, n, shp), dtype=np.float64).reshape(shp)
            for n, shp in meta(case['name'])['variables']}


# ---- TF names <-> oracle/cfl_oracle.py parameter keys ---------------------------------------------------
_HEADS_DIST = {'latent_outputs': 'outputs', 'pcd_outputs': 'proto'}
_HEADS_CFL = {'outputs': 'outputs', 'prototype_outputs': 'proto', 'monomer_outputs': 'mono'}
_LEAF = {'weights': 'W', 'V': 'W', 'biases': 'b', 'g': 'g'}


def split_encoder_name(name):
    """'CFL/DistEncoderSrc/outputs/fully_connected/V' -> ('src', 'outputs/W'); None for other variables"""
    parts = name.split('/')
    if parts[0] == 'Dist' and parts[1] == 'Encoder' and parts[2] in _HEADS_DIST:
        return 'src', _HEADS_DIST[parts[2]] + '/' + _LEAF[parts[-1]]
    if parts[0] == 'CFL' and parts[1].startswith('DistEncoder') and parts[2] in _HEADS_CFL:
        side = 'dst' if parts[1] == 'DistEncoderDst' else 'src'
        return side, _HEADS_CFL[parts[2]] + '/' + _LEAF[parts[-1]]
    return None


def encoder_params(values):
    """(params_src, params_dst | None, raw_threshold) in oracle naming from {TF name: array}"""
    src, dst, thr = {}, {}, None
    for n, v in values.items():
        if n.endswith('Thresholder/threshold/threshold'):
            thr = np.float64(v)
            continue
        sk = split_encoder_name(n)
        if sk is not None:
            (src if sk[0] == 'src' else dst)[sk[1]] = np.asarray(v, dtype=np.float64)
    return src, (dst or None), thr
